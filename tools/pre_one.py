#!/usr/bin/env python3
"""Diagnostic (GPU box): preprocessing of 1 / 64 / 1024 FRESH scans per launch, the literal walk against the whole wave at work."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from laser_slam_amd import capi, synth
raw = bench.build_scan_set(); S0, n = raw.shape
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
for path in ("walk", "wave"):
    os.environ["LSA_PREPROCESS_PATH"] = path
    for Sp in (1, 64, 1024):
        bufs = [(torch.from_numpy(raw[(k * Sp) % (S0 - Sp):(k * Sp) % (S0 - Sp) + Sp].copy()).cuda(), torch.zeros(Sp, n, dtype=torch.uint8, device="cuda"), torch.zeros(Sp, n, dtype=torch.int16, device="cuda")) for k in range(12)]
        ctx.preprocess_batch(Sp, *bufs[0]); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for k in range(1, 11): ctx.preprocess_batch(Sp, *bufs[k])
        b.record(); torch.cuda.synchronize()
        print("%s S=%d: %.1f us per launch (fresh scans)" % (path, Sp, a.elapsed_time(b) * 100), flush=True)
