#!/usr/bin/env python3
"""Diagnostic (GPU box): latency of a lone CICPWarpper::ICPMatch (lsa_p2p_icp_host) on the reference's map-node clouds with their
recorded priors: python3 tools/p2p_lat.py [build_ab/<variant>.so]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.zeros(1, device="cuda")
from laser_slam_amd import capi
from helpers import GOLDEN
if len(sys.argv) > 1: capi.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = capi.Context(capi.param_preset("LMS211"))
def cloud(n):
    a = np.loadtxt(os.path.join(GOLDEN, "zhicp", "%d.log" % n), dtype=np.float32); return a[:, 0].copy(), a[:, 1].copy()
rows = np.loadtxt(os.path.join(GOLDEN, "zhicp", "matched_nodes.log"))
ts, its = [], []
for rep in range(6):
    for n1, n2, _, px, py, pth in rows:
        try: rx, ry = cloud(int(n1)); ax, ay = cloud(int(n2))
        except OSError: continue
        t0 = time.perf_counter(); o = capi.p2p_icp_host(ctx, rx, ry, ax, ay, (px, py, pth)); ts.append(time.perf_counter() - t0); its.append(o[5])
ts = np.array(ts[len(ts)//6:]) * 1e6
print("CICPWarpper::ICPMatch on node clouds (%d..%d points): median %.0f us, p95 %.0f us, %.0f iterations on average" % (len(rx), len(ax), np.median(ts), np.percentile(ts, 95), np.mean(its)))
