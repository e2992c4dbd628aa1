#!/usr/bin/env python3
"""Diagnostic (GPU box): the fused localisation step (C ABI, one call) at the particle counts a robot runs - 60 (the reference's
globaldef.cpp:15 default), 1 024, 16 384 - where launches and the read-back, not the scorer, are the step."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from laser_slam_amd import capi, synth
raw = bench.build_scan_set()
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
grid, fld, d_sx, d_sy, n = bench.pf_world(torch, capi, ctx, raw, p)
for P in (60, 1024, 16384):
    rng = np.random.default_rng(1)
    pp = np.stack([10 + rng.normal(0, 1.0, P), 7.5 + rng.normal(0, 1.0, P), rng.normal(0, 0.3, P)], 1).astype(np.float32)
    d_cur = torch.from_numpy(pp).cuda(); d_out = torch.zeros(P, 3, device="cuda"); d_nrm = torch.randn(P, 3, device="cuda")
    flt = capi.PfFilter(fld, P)
    ts = []
    for k in range(60):
        t0 = time.perf_counter()
        flt.step(d_sx, d_sy, n, 3, d_cur, d_out, P, P, 0.37, resample_threshold=2.0, roughen_K=0.2, d_normals_roughen=d_nrm)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts[10:]) * 1e6
    print("fused step at %6d particles: median %.0f us, p95 %.0f us" % (P, np.median(ts), np.percentile(ts, 95)), flush=True)
