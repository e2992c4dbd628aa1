#!/usr/bin/env python3
"""Diagnostic (GPU box): latency of a lone CCanonicalMatcher::FMatch (lsa_plicp_host) over consecutive scans of the logs' heads,
with the share of calls that restart (more than 10 iterations)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.zeros(1, device="cuda")
from laser_slam_amd import capi
if len(sys.argv) > 1 and sys.argv[1] != "-": capi.LIB_PATH = os.path.abspath(sys.argv[1])
for name, preset, fname, tag in (("fr079", "LMS511", "fr079_head.log", "ROBOTLASER1"), ("intel", "LMS211", "intel_head.log", "ROBOTLASER1"),
                                 ("mit", "LMS511", "mit_head.log", "ROBOTLASER1"), ("indoor", "LMS151", "indoor_head.txt", "LMS151")):
    p = capi.param_preset(preset)
    r, _, _, bad, _ = capi.carmen_read(p, os.path.join(ROOT, "tests", "golden", fname), tag)
    ctx = capi.Context(p)
    capi.plicp_host(ctx, r[0], r[1])
    ts, its = [], []
    for k in range(2, min(len(r), 60)):
        t0 = time.perf_counter(); o = capi.plicp_host(ctx, r[k - 1], r[k]); ts.append(time.perf_counter() - t0); its.append(o[5])
    ts = np.array(ts) * 1e6
    print("%-7s %4d beams  lone FMatch (zhcsm): median %.0f us, p95 %.0f us, max %.0f us; %.0f %% of the calls restart" % (name, r.shape[1], np.median(ts), np.percentile(ts, 95), ts.max(), 100.0 * np.mean(np.array(its) > 10)), flush=True)
    ctx.close()
