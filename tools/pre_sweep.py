#!/usr/bin/env python3
"""Diagnostic (GPU box): the four preprocessing kernels bit-compared on ragged batches, and their rates by batch size (fresh scans
for every launch would change little here: the timed launches re-filter filtered scans, which the sweeps of the wave kernel settle
faster - tools numbers for it are upper bounds; bench.py and the latency entry use fresh scans)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from laser_slam_amd import capi, synth
if len(sys.argv) > 1: capi.LIB_PATH = os.path.abspath(sys.argv[1])
raw = bench.build_scan_set(); S0, n = raw.shape
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
# bit comparison of the three paths on ragged batches of three sensor sizes
rng = np.random.default_rng(5)
for nn, Sp in ((1081, 4999), (361, 777), (181, 64 * 9 + 1), (97, 130)):
    pp = capi.param_make(**dict(synth.SYN1081, points=nn, window=min(synth.SYN1081["window"], (nn - 1) // 2))) if nn != 1081 else p
    cx = capi.Context(pp, 0); cx.set_stream(stream.cuda_stream)
    base = np.tile(raw, ((Sp + S0 - 1) // S0, 1))[:Sp, :nn].copy()
    base += rng.normal(0, 3.0, base.shape).astype(np.float32) * (rng.random(base.shape) < 0.05)
    base[rng.random(base.shape) < 0.01] = 9000.0
    flags = (rng.random(base.shape) < 0.03).astype(np.uint8)
    outs = {}
    for path in ("walk", "wave", "lanes", "tiles"):
        os.environ["LSA_PREPROCESS_PATH"] = path
        r2 = torch.from_numpy(base).cuda(); b2 = torch.from_numpy(flags).cuda(); s2 = torch.zeros(Sp, nn, dtype=torch.int16, device="cuda")
        cx.preprocess_batch(Sp, r2, b2, s2); cx.sync()
        outs[path] = (r2.cpu().numpy().view(np.uint32), b2.cpu().numpy(), s2.cpu().numpy())
    for path in ("wave", "lanes", "tiles"):
        same = all(np.array_equal(a, b) for a, b in zip(outs["walk"], outs[path]))
        print("n=%4d S=%5d  %s == walk: %s" % (nn, Sp, path, same), flush=True)
for path, Sp in [(q, w) for q in ("wave", "lanes", "tiles") for w in (1024, 4096, 16384, 65536, 262144, 1048576)]:
    os.environ["LSA_PREPROCESS_PATH"] = path
    big = np.tile(raw, ((Sp + S0 - 1) // S0, 1))[:Sp]
    r2 = torch.from_numpy(big).cuda(); b2 = torch.zeros(Sp, n, dtype=torch.uint8, device="cuda"); s2 = torch.zeros(Sp, n, dtype=torch.int16, device="cuda")
    ms = bench.timed(torch, lambda: ctx.preprocess_batch(Sp, r2, b2, s2), reps=3)
    print(path, "S=%8d: %.2f M scans/s  %.0f GB/s (12 B per beam)" % (Sp, Sp / ms / 1e3, Sp * n * 12 / ms / 1e6), flush=True)
    del r2, b2, s2
