#!/usr/bin/env python3
"""Diagnostic (GPU box): latency of one CPMap::updateMap (lsa_grid_update_host) of a 1081-beam scan into a 4096 x 4096 grid at 5 cm."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
torch.zeros(1, device="cuda")
from laser_slam_amd import capi, synth
import bench
raw = bench.build_scan_set()[:200]; n = raw.shape[1]
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
fi = capi.param_tables(p)[0]
g = capi.Grid(ctx, 4096, 4096, 2048, 2048, 0.0, 0.0, 0.05)
ts = []
for k in range(120):
    r = raw[k] * 0.01
    x = (r * np.cos(fi) + 1.0).astype(np.float32); y = (r * np.sin(fi) + 2.0).astype(np.float32)
    t0 = time.perf_counter()
    g.update_host(x, y, 1.0, 2.0)
    ctx.sync()
    ts.append(time.perf_counter() - t0)
ts = np.array(ts[20:]) * 1e6
print("updateMap of one 1081-beam scan into 4096^2 @ 5 cm: median %.0f us, p95 %.0f us" % (np.median(ts), np.percentile(ts, 95)))
