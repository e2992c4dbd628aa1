#!/usr/bin/env python3
"""Diagnostic: ICP throughput of a prebuilt variant on 541-beam scans (LDS no longer limits occupancy there,
so builds with different register budgets show how throughput scales with workgroups per CU)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import capi, synth
capi.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
raw = np.concatenate([synth.make_scans(200, n_beams=541, noise_seed=12345 + j)[0] for j in range(4)], 0)
p = capi.param_make(points=541, fov_deg=270, max_range=3000, min_valid=150, window=100, fi_min_deg=-135, fi_max_deg=135)
ctx = capi.Context(p, 0)
S, n = raw.shape
d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
torch.cuda.synchronize(); ctx.preprocess_batch(S, d_r, d_bad, d_seg)
B = 32768
idx = np.arange(B) % (S - 1); idx = np.where(idx % 200 == 199, idx - 1, idx)
ri = torch.from_numpy(idx.astype(np.int32)).cuda(); ai = ri + 1
pose = torch.zeros(B, 3, device="cuda"); it = torch.zeros(B, dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
W = int(os.environ.get("LSA_WHICH", "1"))
ctx.match_batch(W, B, d_r, d_bad, d_seg, ri, ai, pose, d_iters=it); ctx.sync()
t0 = time.perf_counter()
for _ in range(3): ctx.match_batch(W, B, d_r, d_bad, d_seg, ri, ai, pose)
ctx.sync(); dt = (time.perf_counter() - t0) / 3
print("%s 541 beams: %.0f matches/s, mean iterations %.1f" % (os.path.basename(sys.argv[1]), B / dt, float(it.float().mean())))
