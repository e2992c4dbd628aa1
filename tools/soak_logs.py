#!/usr/bin/env python3
"""One-off soak on the shipped logs' heads (tests/golden/*_head.*): every pair up to 6 scans apart, random
priors, both matchers, device against the CPU oracle."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from laser_slam_amd import capi
if os.environ.get("LSA_SOAK_LIB"):          # an older build: tolerate symbols it does not have yet
    import ctypes as _C
    class _Tolerant(_C.CDLL):
        def __getattr__(self, name):
            try:
                return super().__getattr__(name)
            except AttributeError:
                class _Dummy: pass
                return _Dummy()
    capi.C.CDLL = _Tolerant
    capi.LIB_PATH = os.path.abspath(os.environ["LSA_SOAK_LIB"])
from helpers import GOLDEN, SETS, golden_param, load
from oracle_lib import Oracle
rng = np.random.default_rng(3)
for name in ("mit", "fr079", "intel", "indoor"):
    laser, fn, tag = SETS[name]
    g = load(name); gp = golden_param(g)
    o = Oracle(laser); o.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
    p = capi.param_preset(laser); p.fi_min = float(np.float32(gp["fi_min"])); p.fi_max = float(np.float32(gp["fi_max"])); p.max_range = int(gp["max_range"])
    ctx = capi.Context(p)
    r, bad, seg = (np.ascontiguousarray(g[k]) for k in ("pre_r", "pre_bad", "pre_seg"))
    S = r.shape[0]
    pairs = [(i, j) for i in range(S) for j in range(max(0, i - 6), min(S, i + 7))]
    ri = np.array([a for a, _ in pairs], np.int32); ai = np.array([b for _, b in pairs], np.int32)
    B = len(pairs)
    init = np.zeros((B, 6), np.float32)
    init[:, 0:2] = rng.uniform(-200, 200, (B, 2)); init[:, 2] = rng.uniform(-3, 3, B)
    init[:, 3:5] = init[:, 0:2] + rng.uniform(-40, 40, (B, 2)); init[:, 5] = init[:, 2] + rng.uniform(-0.15, 0.15, B)
    init[::3] = 0
    d = lambda x, t: torch.from_numpy(np.ascontiguousarray(x).astype(t)).cuda()
    d_r, d_bad, d_seg, d_ri, d_ai, d_init = d(r, np.float32), d(bad, np.uint8), d(seg, np.int16), d(ri, np.int32), d(ai, np.int32), d(init, np.float32)
    for which in (0, 1):
        pose = torch.zeros(B, 3, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        ctx.match_batch(which, B, d_r, d_bad, d_seg, d_ri, d_ai, pose, d_init, d_status=st); ctx.sync()
        gp_, gs = pose.cpu().numpy(), st.cpu().numpy()
        bads = 0
        for k in range(B):
            s_, p_, _, _ = o.match(which, (r[ri[k]], bad[ri[k]], seg[ri[k]]), (r[ai[k]], bad[ai[k]], seg[ai[k]]), init[k, :3], init[k, 3:])
            if (s_ != gs[k]) or (s_ == 0 and p_.tobytes() != gp_[k].tobytes()):
                bads += 1
                if bads <= 4:
                    print("   ", name, which, "pair", ri[k], ai[k], "init", init[k], "oracle", s_, p_, "device", gs[k], gp_[k], flush=True)
        print("%-7s %s: %d pairs, %d matcher failures, %d mismatches" % (name, "psm" if which == 0 else "icp", B, int((gs != 0).sum()), bads), flush=True)
    ctx.close()
