#!/usr/bin/env python3
"""Diagnostic (GPU box): the likelihood-field scorer by lanes per pose, and one localisation step - fused (the C ABI's
one call; pf.py's phase-by-phase form) against the call-by-call step of round 2 - at 2^20 particles and at the 131 072
one rank of eight would own.  Same map, scan and particle cloud as bench.py's pf entries."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from laser_slam_amd import capi, synth, pf

raw = bench.build_scan_set()
p = capi.param_make(**synth.SYN1081)
ctx = capi.Context(p, 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
grid, fld, d_sx, d_sy, n = bench.pf_world(torch, capi, ctx, raw, p)

def timed(fn, reps=20):
    """median over `reps` of 5 back-to-back calls each (one slow call - an allocation, a page fault - does not show)"""
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
    return float(np.median(ts))

for P in (1 << 20, 1 << 17):
    rng = np.random.default_rng(1)
    pp = np.stack([10 + rng.normal(0, 1.0, P), 7.5 + rng.normal(0, 1.0, P), rng.normal(0, 0.3, P)], 1).astype(np.float32)
    d_p = torch.from_numpy(pp).cuda(); d_sc = torch.zeros(P, device="cuda")
    for L in (8, 16, 32, 64):
        os.environ["LSA_PF_LANES"] = str(L)
        ms = timed(lambda: fld.score_poses(d_sx, d_sy, n, 3, d_p, P, d_sc))
        print("score P=%d L=%d: %.4f ms  %.3f G poses/s" % (P, L, ms, P / ms / 1e6), flush=True)
    os.environ.pop("LSA_PF_LANES")
    ms = timed(lambda: fld.score_poses(d_sx, d_sy, n, 3, d_p, P, d_sc))
    print("score P=%d default: %.4f ms  %.3f G poses/s" % (P, ms, P / ms / 1e6), flush=True)
    d_nrm = torch.randn(P, 3, device="cuda"); d_cur = d_p.clone(); d_out = torch.zeros(P, 3, device="cuda")
    flt = capi.PfFilter(fld, P)
    pred = ("predict", [10.0, 7.5, 0.0], [0.25, 0.25, 0.27])
    ms = timed(lambda: flt.step(d_sx, d_sy, n, 3, d_cur, d_out, P, P, 0.37, resample_threshold=2.0, roughen_K=0.2,
                                d_normals_roughen=d_nrm, predict=pred, d_normals_predict=d_nrm))
    print("step  P=%d fused, one C call: %.4f ms" % (P, ms), flush=True)
    fp = pf.FusedDevicePrims(ctx, fld, P, d_sx, d_sy, n, 3)
    ff = pf.FusedParticleFilter(P, fp, resample_threshold=2.0, roughen_K=0.2)
    ms = timed(lambda: ff.step(d_cur, 0.37, predict=pred, normals_predict=d_nrm, normals_roughen=d_nrm))
    print("step  P=%d fused, pf.py phases: %.4f ms" % (P, ms), flush=True)
    prims = pf.DevicePrims(ctx, fld, d_sx, d_sy, n, 3)
    old = pf.ShardedParticleFilter(P, prims, resample_threshold=2.0, device="cpu")
    def old_step():
        capi.pf_predict(ctx, P, pred[1], pred[2], d_nrm, d_cur)
        new, info = old.step(d_cur, 0.37)
        capi.pf_roughen(ctx, P, 0.2, d_nrm, new); ctx.sync()
    ms = timed(old_step)
    print("step  P=%d call by call (round 2): %.4f ms" % (P, ms), flush=True)
    flt.close(); fp.close()
