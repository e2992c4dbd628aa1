#!/usr/bin/env python3
"""One-off soak of the point-to-point ICP (CICPWarpper's arithmetic) against oracle/p2p_oracle.c: clouds of
synthetic scans up to 10 apart, priors off the true motion by up to 0.3 m / 0.2 rad, thinned / empty clouds."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
torch.zeros(1, device="cuda")
from laser_slam_amd import capi, synth
from oracle_lib import p2p_icp
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
n_beams, fov = 361, 270.0
raw, poses = synth.make_scans(60, n_beams=n_beams, fov_deg=fov)
bear = np.deg2rad(-fov / 2 + fov * np.arange(n_beams) / (n_beams - 1))
ctx = capi.Context(capi.param_preset("LMS211"))
cases = []
for k in range(B):
    a = int(rng.integers(0, 60)); b = int(np.clip(a + rng.integers(-10, 11), 0, 59))
    def cl(i):
        r = raw[i] / 100.0; ok = (raw[i] < 2999) & (rng.uniform(size=n_beams) < rng.choice([1.0, 0.5, 0.1]))
        return (r * np.cos(bear))[ok].astype(np.float32), (r * np.sin(bear))[ok].astype(np.float32)
    rx, ry = cl(a); cx, cy = cl(b)
    if k % 97 == 0: cx, cy = cx[:0], cy[:0]
    x0, y0, t0 = poses[a]; x1, y1, t1 = poses[b]
    c, s = np.cos(-t0), np.sin(-t0)
    init = np.array([c * (x1 - x0) - s * (y1 - y0), s * (x1 - x0) + c * (y1 - y0), t1 - t0]) + rng.uniform(-1, 1, 3) * [0.3, 0.3, 0.2] * rng.choice([0.1, 1.0])
    cases.append((rx, ry, cx, cy, init))
got = [capi.p2p_icp_host(ctx, *c_) for c_ in cases]
def work(ks):
    out = []
    for k in ks:
        w = p2p_icp(*cases[k]); g = got[k]
        if g[5] != w[5] or g[6] != w[6] or not np.allclose(g[:5], w[:5], rtol=1e-9, atol=1e-9) or not np.allclose(g[7:13], w[7:13], rtol=1e-6, atol=1e-12):
            out.append((k, w, g))
    return out
t0 = time.time()
with ThreadPoolExecutor(16) as ex:
    bads = sum(ex.map(work, np.array_split(np.arange(B), 16)), [])
print("p2p icp: %d pairs, %d mismatches (%.0f s)" % (B, len(bads), time.time() - t0))
np.set_printoptions(linewidth=220)
for k, w, g in bads[:5]:
    print("  case", k, "sizes", len(cases[k][0]), len(cases[k][2]), "\n   oracle", w[:13], "\n   device", g[:13])
