#!/usr/bin/env python3
"""Diagnostic (GPU box): latency of a lone CPolarMatch::pm_psm / pm_icp (lsa_match_host on preprocessed scans) over consecutive
scans of the logs' heads."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
torch.zeros(1, device="cuda")
from laser_slam_amd import capi
if len(sys.argv) > 1 and sys.argv[1] != "-": capi.LIB_PATH = os.path.abspath(sys.argv[1])
for name, preset, fname, tag in (("fr079", "LMS511", "fr079_head.log", "ROBOTLASER1"), ("intel", "LMS211", "intel_head.log", "ROBOTLASER1"), ("indoor", "LMS151", "indoor_head.txt", "LMS151")):
    p = capi.param_preset(preset)
    r, _, _, bad, _ = capi.carmen_read(p, os.path.join(ROOT, "tests", "golden", fname), tag)
    ctx = capi.Context(p)
    S, n = r.shape
    pre = [ctx.preprocess_host(r[k], bad[k].astype(np.int32)) for k in range(min(S, 60))]
    for which, nm in ((0, "pm_psm"), (1, "pm_icp")):
        ts = []
        for k in range(1, len(pre)):
            t0 = time.perf_counter(); ctx.match_host(which, pre[k - 1], pre[k]); ts.append(time.perf_counter() - t0)
        ts = np.array(ts[5:]) * 1e6
        print("%-7s %4d beams  lone %s: median %.0f us, p95 %.0f us" % (name, n, nm, np.median(ts), np.percentile(ts, 95)), flush=True)
    ctx.close()
