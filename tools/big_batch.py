#!/usr/bin/env python3
"""Diagnostic: one launch of 2^20 pairs per matcher (SURVEY 8(d) batch sizes): rate, and every copy of a pair must carry the same bits."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import capi, synth
import torch
raw = np.concatenate([synth.make_scans(200, noise_seed=12345 + j)[0] for j in range(2)], 0)
S, n = raw.shape
ctx = capi.Context(capi.param_make(**synth.SYN1081))
d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
torch.cuda.synchronize()
ctx.preprocess_batch(S, d_r, d_bad, d_seg); ctx.sync()
base = np.array([(s * 200 + k - 1, s * 200 + k) for s in range(2) for k in range(1, 200)], np.int32)
B = 1 << 20
sel = np.arange(B) % len(base)
ri = torch.from_numpy(np.ascontiguousarray(base[sel, 0])).cuda(); ai = torch.from_numpy(np.ascontiguousarray(base[sel, 1])).cuda()
for which, nm in ((1, "icp"), (0, "psm")):
    pose = torch.zeros(B, 3, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ctx.match_batch(which, B, d_r, d_bad, d_seg, ri, ai, pose, d_status=st); ctx.sync()
    dt = time.perf_counter() - t0
    p = pose.cpu().numpy().view(np.uint32).reshape(B, 3); s_ = st.cpu().numpy()
    k = len(base)
    full = (B // k) * k
    same = np.array_equal(p[:full].reshape(-1, k, 3), np.broadcast_to(p[:k], (B // k, k, 3))) and np.array_equal(s_[:full].reshape(-1, k), np.broadcast_to(s_[:k], (B // k, k)))
    print("pm_%s: 2^20 pairs in %.2f s = %.0f matches/s, failures %.3f, all copies identical: %s" % (nm, dt, B / dt, float((s_ != 0).mean()), same))
