#!/usr/bin/env python3
"""One-off soak of the point-to-line ICP against oracle/plicp_oracle.c: every pair of the shipped logs' heads
up to 6 scans apart plus random synthetic pairs up to 12 apart. The device sums its normal equations in another
order than the host, so records are compared within 5e-6 (pose), the validity flag exactly."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from laser_slam_amd import capi, synth
from helpers import SETS, golden_param, load
from oracle_lib import plicp
torch.zeros(1, device="cuda")
if os.environ.get("LSA_SOAK_LIB"):
    capi.LIB_PATH = os.path.abspath(os.environ["LSA_SOAK_LIB"])

def run(tag, p, raw, ri, ai):
    ctx = capi.Context(p)
    B = len(ri)
    d_out = torch.zeros(B, 8, dtype=torch.float64, device="cuda")
    d_r = torch.from_numpy(np.ascontiguousarray(raw, np.float32)).cuda()
    torch.cuda.synchronize()
    capi.plicp_batch(ctx, B, d_r, torch.from_numpy(ri).cuda(), torch.from_numpy(ai).cuda(), d_out); ctx.sync()
    got = d_out.cpu().numpy()
    def work(ks):
        out = []
        for k in ks:
            w = plicp(raw[ri[k]], raw[ai[k]], p.fi_min, p.dfi, p.min_range, p.max_range)
            g = got[k]
            if w[3] != g[3] or (w[3] == 1.0 and (np.abs(w[:3] - g[:3]).max() > 5e-6 or w[5] != g[5])):
                out.append((k, w, g))
        return out
    t0 = time.time()
    with ThreadPoolExecutor(16) as ex:
        bads = sum(ex.map(work, np.array_split(np.arange(B), 16)), [])
    print("%-8s plicp: %d pairs, %d invalid, %d mismatches (%.0f s)" % (tag, B, int((got[:, 3] != 1.0).sum()), len(bads), time.time() - t0), flush=True)
    for k, w, g in bads[:4]:
        print("    pair", ri[k], ai[k], "oracle", w, "device", g, flush=True)
    ctx.close()

for name in ("mit", "fr079", "intel", "indoor"):
    laser, fn, tag = SETS[name]
    g = load(name); gp = golden_param(g)
    p = capi.param_preset(laser); p.fi_min = float(np.float32(gp["fi_min"])); p.fi_max = float(np.float32(gp["fi_max"])); p.max_range = int(gp["max_range"])
    raw = np.ascontiguousarray(g["raw_r"], np.float32); S = raw.shape[0]
    pairs = [(i, j) for i in range(S) for j in range(max(0, i - 6), min(S, i + 7))]
    run(name, p, raw, np.array([a for a, _ in pairs], np.int32), np.array([b for _, b in pairs], np.int32))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
rng = np.random.default_rng(11)
raw = np.concatenate([synth.make_scans(60, noise_seed=synth.WORLD_SEED + 700 + j, sigma_m=(0.01 if j < 2 else 0.03))[0] for j in range(4)], 0)
base = rng.integers(0, 4, B) * 60
a = rng.integers(0, 60, B); b = np.clip(a + rng.integers(-12, 13, B), 0, 59)
run("syn1081", capi.param_make(**synth.SYN1081), raw, (base + a).astype(np.int32), (base + b).astype(np.int32))
