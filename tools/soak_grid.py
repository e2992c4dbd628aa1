#!/usr/bin/env python3
"""One-off soak of the batched grid update (owned-tile path) against the oracle: random map geometries
(sizes that are no multiples of the tile, shifted centres / offsets, coarse and fine cells), robots inside and
outside the map, rays ending far outside, ragged scans."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from laser_slam_amd import capi
from oracle_lib import OraclePMap
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ctx = capi.Context(capi.param_preset("LMS151"))
bad = 0
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    sx, sy = int(rng.integers(60, 700)), int(rng.integers(60, 700))
    cx, cy = int(rng.integers(0, sx)), int(rng.integers(0, sy))
    offx, offy = float(np.float32(rng.uniform(-5, 5))), float(np.float32(rng.uniform(-5, 5)))
    res = float(rng.choice([0.02, 0.05, 0.1, 0.25]))
    S, nb = int(rng.integers(8, 120)), int(rng.integers(20, 700))
    ext = res * max(sx, sy)
    rob = np.stack([offx + rng.uniform(-0.7, 0.7, S) * ext, offy + rng.uniform(-0.7, 0.7, S) * ext], 1).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, (S, nb)); d = rng.uniform(0, 1, (S, nb)) ** 2 * ext * rng.choice([0.3, 1.0, 3.0])
    gx = (rob[:, :1] + d * np.cos(ang)).astype(np.float32); gy = (rob[:, 1:] + d * np.sin(ang)).astype(np.float32)
    nv = rng.integers(0, nb + 1, S).astype(np.int32) if trial % 2 else None
    grid = capi.Grid(ctx, sx, sy, cx, cy, offx, offy, res)
    o = OraclePMap(sx, sy, cx, cy, offx, offy, res)
    for s in range(S):
        n = nb if nv is None else int(nv[s])
        if n:
            o.update(gx[s, :n], gy[s, :n], float(rob[s, 0]), float(rob[s, 1]))
    d_ = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    torch.cuda.synchronize()
    grid.update_batch(S, nb, d_(gx), d_(gy), d_(rob), None if nv is None else d_(nv))
    grid.compute_prob(); o.compute()
    ok = all(np.array_equal(a.view(np.uint32), b.view(np.uint32)) for a, b in zip(grid.layers(), o.layers()))
    if not ok:
        bad += 1
        print("MISMATCH trial", trial, dict(sx=sx, sy=sy, cx=cx, cy=cy, offx=offx, offy=offy, res=res, S=S, nb=nb, ragged=nv is not None), flush=True)
    grid.close()
print("grid soak: %d trials, %d mismatching" % (trial + 1, bad))
