#!/usr/bin/env python3
"""Diagnostic: preprocessing rate of the two kernels by batch size, and bit comparison of their outputs."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import capi, synth
import torch
raw = np.concatenate([synth.make_scans(200, noise_seed=12345 + j)[0] for j in range(4)], 0)
raw[5, 100:140] = 3001.0; raw[7, ::9] = 5.0
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p)
n = raw.shape[1]
for S in (256, 1024, 4096, 16384, 65536):
    big = np.tile(raw, ((S + len(raw) - 1) // len(raw), 1))[:S]
    outs = {}
    for path in ("wave", "lanes"):
        os.environ["LSA_PREPROCESS_PATH"] = path
        d_r = torch.from_numpy(big).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        ctx.preprocess_batch(S, d_r, d_bad, d_seg); ctx.sync()
        outs[path] = (d_r.cpu().numpy().copy(), d_bad.cpu().numpy().copy(), d_seg.cpu().numpy().copy())
        d_r2 = torch.from_numpy(big).cuda(); torch.cuda.synchronize()
        t0 = time.perf_counter(); ctx.preprocess_batch(S, d_r2, d_bad, d_seg); ctx.sync(); dt = time.perf_counter() - t0
        outs[path + "_rate"] = S / dt
    same = all(np.array_equal(a.view(np.uint8), b.view(np.uint8)) for a, b in zip(outs["wave"], outs["lanes"]))
    print("S=%6d  wave-per-scan %.2f M scans/s   lane-per-scan %.2f M scans/s   identical: %s" % (S, outs["wave_rate"] / 1e6, outs["lanes_rate"] / 1e6, same))
