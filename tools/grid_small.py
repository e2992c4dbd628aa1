#!/usr/bin/env python3
"""Diagnostic: grid-update time per batch for small batches, binned path against the per-scan kernel."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import capi, synth
import torch
raw, _ = synth.make_scans(256)
p = capi.param_make(**synth.SYN1081)
fi = capi.param_tables(p)[0]
ctx = capi.Context(p)
g = capi.Grid(ctx, 4096, 4096, 1024, 1024, 0.0, 0.0, 0.05)
for nb in (8, 16, 32, 64, 128, 256, 512):
    poses = synth.path_pose(np.arange(nb))
    rr = raw[:nb] / 100.0 if nb <= 256 else np.concatenate([raw] * 2)[:nb] / 100.0
    gx = (poses[:, 0:1] + rr * np.cos(poses[:, 2:3] + fi[None])).astype(np.float32)
    gy = (poses[:, 1:2] + rr * np.sin(poses[:, 2:3] + fi[None])).astype(np.float32)
    dgx = torch.from_numpy(gx).cuda(); dgy = torch.from_numpy(gy).cuda(); drob = torch.from_numpy(poses[:, :2].astype(np.float32)).cuda()
    out = []
    for path in ("", "scan"):
        if path: os.environ["LSA_GRID_PATH"] = path
        else: os.environ.pop("LSA_GRID_PATH", None)
        g.update_batch(nb, 1081, dgx, dgy, drob); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(10): g.update_batch(nb, 1081, dgx, dgy, drob)
        ctx.sync(); out.append((time.perf_counter() - t0) / 10 * 1e6)
    print("S=%4d  binned %7.1f us   per-scan kernel %7.1f us" % (nb, out[0], out[1]))
