#!/usr/bin/env python3
"""Diagnostic (GPU box): where the point-to-line ICP's workgroups spend their cycles, per phase of an iteration (a
-DLSA_PLICP_STAMPS build; thread 0's clock): window search, doubles test, distances + trimming (two order statistics),
normal-equation sums, closed-form step, hash / loop test.  usage: plicp_phases.py [pairs] [fr079|intel|mit|indoor]"""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
out = os.path.join(ROOT, "gpurun_out", "liblsa_core_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc"] + ge.HIP_FLAGS + ["-DLSA_PLICP_STAMPS", "-o", out, os.path.join(ge.CSRC, "core.hip"), os.path.join(ge.CSRC, "carmen.cpp")])
from laser_slam_amd import capi, synth
capi.LIB_PATH = out
import torch
import bench
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
if len(sys.argv) > 2:
    name = sys.argv[2]
    preset, fname, tag = {"fr079": ("LMS511", "fr079_head.log", "ROBOTLASER1"), "intel": ("LMS211", "intel_head.log", "ROBOTLASER1"),
                          "mit": ("LMS511", "mit_head.log", "ROBOTLASER1"), "indoor": ("LMS151", "indoor_head.txt", "LMS151")}[name]
    p = capi.param_preset(preset)
    raw = capi.carmen_read(p, os.path.join(ROOT, "tests", "golden", fname), tag)[0]
else:
    p = capi.param_make(**synth.SYN1081); raw = bench.build_scan_set()[:400]
ctx = capi.Context(p, 0)
S, n = raw.shape
d_r = torch.from_numpy(np.ascontiguousarray(raw)).cuda()
idx = (np.arange(B) % (S - 1)).astype(np.int32)
ri = torch.from_numpy(idx).cuda(); ai = ri + 1
d_out = torch.zeros(B, 8, dtype=torch.float64, device="cuda")
buf = (C.c_ulonglong * 8)()
capi.plicp_batch(ctx, B, d_r, ri, ai, d_out); ctx.sync()
capi.lib().lsa_debug_plicp_cycles(buf)
capi.plicp_batch(ctx, B, d_r, ri, ai, d_out); ctx.sync()
capi.lib().lsa_debug_plicp_cycles(buf)
v = np.array(list(buf), np.float64); its = float(d_out[:, 5].sum())
names = ["window search", "doubles", "distances + trim", "sums (14)", "closed form", "hash / loop", "search: set-up", "search: scan"]
for nm, x in zip(names, v): print("%-18s %6.2f %%  %8.0f cycles per iteration" % (nm, 100 * x / v[:6].sum(), x / its))
print("cycles per iteration %.0f, iterations per pair %.2f" % (v[:6].sum() / its, its / B))
