#!/usr/bin/env python3
"""Diagnostic (GPU box): the likelihood-field scorer alone, `reps` launches over P poses x 361 beams on bench.py's map -
the target of the PMC passes behind profiles/r03_pmc_score_poses.json (usage: score_probe.py [P] [reps])."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from laser_slam_amd import capi, synth

P = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
raw = bench.build_scan_set()
p = capi.param_make(**synth.SYN1081)
ctx = capi.Context(p, 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
grid, fld, d_sx, d_sy, n = bench.pf_world(torch, capi, ctx, raw, p)
d_p = bench.pf_cloud(torch, P, 1); d_sc = torch.zeros(P, device="cuda")
fld.score_poses(d_sx, d_sy, n, 3, d_p, P, d_sc); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    fld.score_poses(d_sx, d_sy, n, 3, d_p, P, d_sc)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / reps * 1e3
print("score_poses P=%d: %.4f ms  %.3f G poses/s  (L=%s)" % (P, ms, P / ms / 1e6, os.environ.get("LSA_PF_LANES", "default")))
