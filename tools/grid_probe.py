#!/usr/bin/env python3
"""Diagnostic: grid-update throughput (2048 synthetic 1081-beam scans into a 4096^2 grid) of a prebuilt variant."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import capi, synth
capi.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
raw, _ = synth.make_scans(256)
p = capi.param_make(**synth.SYN1081)
fi = capi.param_tables(p)[0]
nb = 2048
poses = synth.path_pose(np.arange(nb))
rr = np.concatenate([raw] * 8)[:nb] / 100.0
gx = (poses[:, 0:1] + rr * np.cos(poses[:, 2:3] + fi[None])).astype(np.float32)
gy = (poses[:, 1:2] + rr * np.sin(poses[:, 2:3] + fi[None])).astype(np.float32)
ctx = capi.Context(p)
g = capi.Grid(ctx, 4096, 4096, 1024, 1024, 0.0, 0.0, 0.05)
dgx = torch.from_numpy(gx).cuda(); dgy = torch.from_numpy(gy).cuda(); drob = torch.from_numpy(poses[:, :2].astype(np.float32)).cuda()
torch.cuda.synchronize()
g.update_batch(nb, 1081, dgx, dgy, drob); ctx.sync()
t0 = time.perf_counter()
for _ in range(3): g.update_batch(nb, 1081, dgx, dgy, drob)
ctx.sync(); dt = (time.perf_counter() - t0) / 3
print("%s splits=%s: %.0f scans/s (%.2f ms per 2048 scans)" % (os.path.basename(sys.argv[1]), os.environ.get("LSA_GRID_SPLITS", "auto"), nb / dt, dt * 1e3))

if hasattr(capi.lib(), "lsa_debug_grid_cycles") or True:
    import ctypes as C
    try:
        f = capi.lib().lsa_debug_grid_cycles
    except AttributeError:
        f = None
    if f is not None:
        buf = (C.c_ulonglong * 16)()
        f(buf)                                              # reset
        g.update_batch(nb, 1081, dgx, dgy, drob); ctx.sync()
        f(buf)
        v = list(buf)
        tot = float(v[7]) or float(v[0] + v[1] + v[2] + v[3] + v[4]) or 1.0
        if v[13]:
            print("binned walk: entries-with-crossing %d, lane spans %d (%.1f per crossing lane-slot of 64), wave trip counts %d (mean %.1f), cells added %d -> lane utilisation %.1f%% of wave-iterations x 64"
                  % (v[13], v[10], v[10] / (64.0 * v[13]), v[11], v[11] / float(v[13]), v[12], 100.0 * v[12] / (64.0 * v[11])))
        print("workgroups %d items %d blocks walked %d | share of workgroup time: cull %.1f%% walk %.1f%% zero+tail %.1f%% flush %.1f%% claim/idle %.1f%% | longest item %.0f cycles, mean workgroup life %.0f cycles"
              % (v[9], v[5], v[8], 100 * v[1] / tot, 100 * v[2] / tot, 100 * v[0] / tot, 100 * v[3] / tot, 100 * v[4] / tot, v[6], tot / max(v[9], 1)))
