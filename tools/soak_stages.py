#!/usr/bin/env python3
"""One-off soak of the stand-alone stage kernels against the oracle on the logs' heads and the synthetic set:
pm_scan_project with random poses up to +-pi / +-6 m, pm_error_index2 on pairs up to 6 apart with random priors."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.zeros(1, device="cuda")
from laser_slam_amd import capi
from helpers import SETS, golden_param, load
from oracle_lib import Oracle
rng = np.random.default_rng(17)
d = lambda x, t: torch.from_numpy(np.ascontiguousarray(x).astype(t)).cuda()
for name in list(SETS):
    laser, fn, tag = SETS[name]
    g = load(name); gp = golden_param(g)
    if name == "syn1081":
        from oracle_lib import SYN1081
        from laser_slam_amd import synth
        o = Oracle(**SYN1081); p = capi.param_make(**synth.SYN1081)
    else:
        o = Oracle(laser); o.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
        p = capi.param_preset(laser); p.fi_min = float(np.float32(gp["fi_min"])); p.fi_max = float(np.float32(gp["fi_max"])); p.max_range = int(gp["max_range"])
    ctx = capi.Context(p)
    r, bad, seg = (np.ascontiguousarray(g[k]) for k in ("pre_r", "pre_bad", "pre_seg"))
    S, n = r.shape
    # scan projection: every scan x 12 random poses
    K = 12
    idx = np.repeat(np.arange(S), K)
    pose = np.stack([rng.uniform(-600, 600, S * K), rng.uniform(-600, 600, S * K), rng.uniform(-3.3, 3.3, S * K)], 1).astype(np.float32)
    pose[::5, :2] *= 0.02; pose[::7, 2] *= 0.01
    nr = torch.zeros(S * K, n, device="cuda"); nb = torch.zeros(S * K, n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.scan_project_batch(S * K, d(r[idx], np.float32), d(bad[idx], np.uint8), d(seg[idx], np.int16), d(pose, np.float32), nr, nb); ctx.sync()
    gr, gb = nr.cpu().numpy(), nb.cpu().numpy().astype(np.int32)
    badp = 0
    for k in range(S * K):
        wr, wb = o.scan_project(r[idx[k]], bad[idx[k]], seg[idx[k]], pose[k])
        if wr.tobytes() != gr[k].tobytes() or not np.array_equal(wb, gb[k]):
            badp += 1
            if badp <= 3: print("   project mismatch", name, idx[k], pose[k], int((wr.view(np.uint32) != gr[k].view(np.uint32)).sum()), int((wb != gb[k]).sum()))
    # error index: pairs up to 6 apart, random priors
    pairs = [(i, j) for i in range(S) for j in range(max(0, i - 6), min(S, i + 7))]
    ri = np.array([a for a, _ in pairs], np.int32); ai = np.array([b for _, b in pairs], np.int32)
    B = len(pairs)
    init = np.zeros((B, 6), np.float32)
    init[:, 0:2] = rng.uniform(-200, 200, (B, 2)); init[:, 2] = rng.uniform(-3, 3, B)
    init[:, 3:5] = init[:, 0:2] + rng.uniform(-60, 60, (B, 2)); init[:, 5] = init[:, 2] + rng.uniform(-0.3, 0.3, B)
    init[::3] = 0
    exy = torch.zeros(B, 2, device="cuda"); na = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.error_index2_batch(B, d(r, np.float32), d(bad, np.uint8), d(seg, np.int16), d(ri, np.int32), d(ai, np.int32), d(init, np.float32), exy, na); ctx.sync()
    ge, gn = exy.cpu().numpy(), na.cpu().numpy()
    bade = 0
    for k in range(B):
        we, wn = o.error_index2((r[ri[k]], bad[ri[k]]), (r[ai[k]], bad[ai[k]], seg[ai[k]]), init[k, :3], init[k, 3:])
        if wn != gn[k] or (wn > 0 and we.tobytes() != ge[k].tobytes()):
            bade += 1
            if bade <= 3: print("   error index mismatch", name, ri[k], ai[k], init[k], we, wn, ge[k], gn[k])
    print("%-8s scan_project: %d cases, %d mismatches | error_index2: %d pairs, %d mismatches" % (name, S * K, badp, B, bade), flush=True)
    ctx.close()
