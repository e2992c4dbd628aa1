#!/usr/bin/env python3
"""Diagnostic: latency of the single-pair host entry points (the reference's sequential FMatch call pattern)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from laser_slam_amd import capi, synth
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
for nb, win, mv in ((181, 20, 40), (361, 40, 80), (541, 50, 100), (1081, 200, 300)):
    kw = dict(points=nb, fov_deg=270 if nb != 181 else 180, max_range=3000, min_valid=mv, window=win,
              fi_min_deg=-135 if nb != 181 else -90, fi_max_deg=135 if nb != 181 else 90)
    ctx = capi.Context(capi.param_make(**kw))
    raw, _ = synth.make_scans(40, n_beams=nb, fov_deg=kw["fov_deg"])
    z = np.zeros(nb, np.int32)
    cur = [raw[0].copy(), z.copy(), z.copy()]
    # warm up
    pose, pair, r2, a2 = ctx.fmatch_host(cur, [raw[1].copy(), z.copy(), z.copy()], True)
    cur = a2
    t0 = time.perf_counter(); K = 38
    for k in range(2, 2 + K):
        pose, pair, r2, a2 = ctx.fmatch_host(cur, [raw[k].copy(), z.copy(), z.copy()], False)
        cur = a2
    dt = (time.perf_counter() - t0) / K
    ref = (raw[0], z, z)
    t0 = time.perf_counter()
    for k in range(K):
        st, p, ret = ctx.match_host(1, (cur[0], cur[1], cur[2]), (cur[0], cur[1], cur[2]))
    dt2 = (time.perf_counter() - t0) / K
    print("%4d beams: lsa_fmatch_host %.0f us per call (%.0f calls/s), lsa_match_host(icp) %.0f us" % (nb, dt * 1e6, 1 / dt, dt2 * 1e6))
    ctx.close()
