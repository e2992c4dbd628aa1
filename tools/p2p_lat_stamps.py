#!/usr/bin/env python3
"""Diagnostic (GPU box): where a lone point-to-point ICP's iteration goes - thread 0's clock per phase (a -DLSA_PHASE_STAMPS build,
P2P_CLOCK in p2p_kernels.hip) on the reference's map-node clouds."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
out = os.path.join(ROOT, "gpurun_out", "liblsa_core_stamps.so")
subprocess.check_call(["hipcc"] + ge.HIP_FLAGS + ["-DLSA_PHASE_STAMPS", "-o", out, os.path.join(ge.CSRC, "core.hip"), os.path.join(ge.CSRC, "carmen.cpp")])
import numpy as np, torch
torch.zeros(1, device="cuda")
from laser_slam_amd import capi
capi.LIB_PATH = out
from helpers import GOLDEN
ctx = capi.Context(capi.param_preset("LMS211"))
def cloud(n):
    a = np.loadtxt(os.path.join(GOLDEN, "zhicp", "%d.log" % n), dtype=np.float32); return a[:, 0].copy(), a[:, 1].copy()
rows = np.loadtxt(os.path.join(GOLDEN, "zhicp", "matched_nodes.log"))
buf = (C.c_ulonglong * 16)(); capi.lib().lsa_debug_phase_cycles(buf)
its = 0
for n1, n2, _, px, py, pth in rows:
    try: rx, ry = cloud(int(n1)); ax, ay = cloud(int(n2))
    except OSError: continue
    o = capi.p2p_icp_host(ctx, rx, ry, ax, ay, (px, py, pth)); its += o[5]
capi.lib().lsa_debug_phase_cycles(buf)
v = list(buf)
print("cycles per iteration: reset %.0f  search %.0f  sums %.0f  closed form %.0f  (%.0f iterations)" % (v[8] / its, v[9] / its, v[10] / its, v[11] / its, its))
