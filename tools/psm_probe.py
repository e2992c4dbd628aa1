#!/usr/bin/env python3
"""Diagnostic (GPU box): pm_psm / pm_icp rate of a build on the synthetic 1081-beam set (2^17 pairs) - for A/B runs of
workgroup widths (usage: psm_probe.py [lib])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from laser_slam_amd import capi, synth
if len(sys.argv) > 1 and sys.argv[1] != "-":
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
raw = bench.build_scan_set(); S, n = raw.shape
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
ctx.preprocess_batch(S, d_r, d_bad, d_seg)
B = 1 << 17
base = np.array([(s * bench.N_BASE + k - 1, s * bench.N_BASE + k) for s in range(bench.SEEDS) for k in range(1, bench.N_BASE)], np.int32)
sel = np.arange(B) % len(base)
ri = torch.from_numpy(np.ascontiguousarray(base[sel, 0])).cuda(); ai = torch.from_numpy(np.ascontiguousarray(base[sel, 1])).cuda()
pose = torch.zeros(B, 3, device="cuda")
for which, nm in ((capi.MATCH_PSM, "pm_psm"), (capi.MATCH_ICP, "pm_icp")):
    ms = bench.timed(torch, lambda: ctx.match_batch(which, B, d_r, d_bad, d_seg, ri, ai, pose), reps=3)
    print("%s 1081 beams: %.3f M matches/s" % (nm, B / ms / 1e3))
