#!/usr/bin/env python3
"""Diagnostic: pm_icp / pm_psm / point-to-line ICP throughput for the reference's sensor sizes (synthetic scans)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import capi, synth
import torch
for n, fov, win, mv in ((181, 180, 20, 40), (361, 180, 40, 80), (541, 270, 50, 100), (1081, 270, 200, 300)):
    raw = np.concatenate([synth.make_scans(200, n_beams=n, fov_deg=float(fov), noise_seed=12345 + j)[0] for j in range(4)], 0)
    p = capi.param_make(points=n, fov_deg=fov, max_range=3000, min_valid=mv, window=win, fi_min_deg=-fov // 2, fi_max_deg=fov // 2)
    ctx = capi.Context(p, 0)
    S = raw.shape[0]
    d_raw = torch.from_numpy(raw).cuda()
    d_r = d_raw.clone(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize(); ctx.preprocess_batch(S, d_r, d_bad, d_seg)
    B = 32768
    idx = np.arange(B) % (S - 1); idx = np.where(idx % 200 == 199, idx - 1, idx)
    ri = torch.from_numpy(idx.astype(np.int32)).cuda(); ai = ri + 1
    pose = torch.zeros(B, 3, device="cuda"); it = torch.zeros(B, dtype=torch.int32, device="cuda")
    o8 = torch.zeros(B, 8, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
    res = []
    for which in (1, 0):
        ctx.match_batch(which, B, d_r, d_bad, d_seg, ri, ai, pose, d_iters=it); ctx.sync()
        t0 = time.perf_counter(); ctx.match_batch(which, B, d_r, d_bad, d_seg, ri, ai, pose); ctx.sync()
        res.append(B / (time.perf_counter() - t0))
    capi.plicp_batch(ctx, 8192, d_raw, ri, ai, o8); ctx.sync()
    t0 = time.perf_counter(); capi.plicp_batch(ctx, 8192, d_raw, ri, ai, o8); ctx.sync()
    res.append(8192 / (time.perf_counter() - t0))
    print("%4d beams (window %3d): pm_icp %.2f M/s, pm_psm %.2f M/s, point-to-line ICP %.2f M/s" % (n, win, res[0] / 1e6, res[1] / 1e6, res[2] / 1e6))
    ctx.close()
