#!/usr/bin/env python3
"""Diagnostic: the matchers' frame-to-frame motion on the shipped logs' heads against the logs' own (corrected) laser
poses - the measure of the reference's testAccuracy (ZHPolar_Match.cpp:399-439).  Accuracy of the ALGORITHMS, the
same on CPU and GPU (their poses carry the same bits)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.zeros(1, device="cuda")
from laser_slam_amd import capi
from helpers import SETS, golden_param, load
for name in ("mit", "fr079", "intel"):
    laser = SETS[name][0]
    g = load(name); gp = golden_param(g)
    p = capi.param_preset(laser); p.fi_min = float(np.float32(gp["fi_min"])); p.fi_max = float(np.float32(gp["fi_max"])); p.max_range = int(gp["max_range"])
    ctx = capi.Context(p)
    r, bad, seg, pose = g["pre_r"], g["pre_bad"], g["pre_seg"], g["reader_pose"].astype(np.float64)
    S = r.shape[0]
    d = lambda x, t: torch.from_numpy(np.ascontiguousarray(x).astype(t)).cuda()
    ri = np.arange(S - 1, dtype=np.int32); ai = ri + 1
    c, s = np.cos(pose[:-1, 2]), np.sin(pose[:-1, 2])
    dxw, dyw = pose[1:, 0] - pose[:-1, 0], pose[1:, 1] - pose[:-1, 1]
    gt = np.stack([c * dxw + s * dyw, -s * dxw + c * dyw, np.arctan2(np.sin(pose[1:, 2] - pose[:-1, 2]), np.cos(pose[1:, 2] - pose[:-1, 2]))], 1)
    line = "%-6s (%d beams, %d pairs, mean motion %.3f m / %.3f rad):" % (name, r.shape[1], S - 1, np.hypot(gt[:, 0], gt[:, 1]).mean(), np.abs(gt[:, 2]).mean())
    for which, nm in ((0, "pm_psm"), (1, "pm_icp")):
        out = torch.zeros(S - 1, 3, device="cuda"); st = torch.zeros(S - 1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.match_batch(which, S - 1, d(r, np.float32), d(bad, np.uint8), d(seg, np.int16), d(ri, np.int32), d(ai, np.int32), out, d_status=st); ctx.sync()
        o, ok = out.cpu().numpy().astype(np.float64), st.cpu().numpy() == 0
        e = np.stack([o[:, 0] * 0.01 - gt[:, 0], o[:, 1] * 0.01 - gt[:, 1], np.arctan2(np.sin(o[:, 2] - gt[:, 2]), np.cos(o[:, 2] - gt[:, 2]))], 1)[ok]
        line += "  %s %d ok, rmse %.3f m / %.4f rad;" % (nm, ok.sum(), np.sqrt((e[:, 0] ** 2 + e[:, 1] ** 2).mean()), np.sqrt((e[:, 2] ** 2).mean()))
    o8 = torch.zeros(S - 1, 8, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
    capi.plicp_batch(ctx, S - 1, d(g["raw_r"], np.float32), d(ri, np.int32), d(ai, np.int32), o8); ctx.sync()
    o = o8.cpu().numpy(); ok = o[:, 3] == 1.0
    e = np.stack([o[:, 0] - gt[:, 0], o[:, 1] - gt[:, 1], np.arctan2(np.sin(o[:, 2] - gt[:, 2]), np.cos(o[:, 2] - gt[:, 2]))], 1)[ok]
    line += "  point-to-line ICP %d ok, rmse %.3f m / %.4f rad" % (ok.sum(), np.sqrt((e[:, 0] ** 2 + e[:, 1] ** 2).mean()), np.sqrt((e[:, 2] ** 2).mean()))
    print(line)
    ctx.close()
