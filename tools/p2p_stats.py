#!/usr/bin/env python3
"""Diagnostic (GPU box): search statistics of p2p_icp_kernel on bench.py's zhicp problems (a -DLSA_PHASE_STAMPS build):
queries, unseeded searches, searches saved by the no-partner cache, four-candidate steps of seeded / unseeded lanes and
steps a wave executed."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
out = os.path.join(ROOT, "gpurun_out", "liblsa_core_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc"] + ge.HIP_FLAGS + ["-DLSA_PHASE_STAMPS", "-o", out, os.path.join(ge.CSRC, "core.hip"), os.path.join(ge.CSRC, "carmen.cpp")])
from laser_slam_amd import capi
capi.LIB_PATH = out
import torch
import bench
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
buf = (C.c_ulonglong * 16)()
orig = bench.p2p_entry
def entry(torch_, capi_, ctx, prob, count, what):
    capi.lib().lsa_debug_phase_cycles(buf)                                  # (reads and clears)
    e = orig(torch_, capi_, ctx, prob, count, what)
    capi.lib().lsa_debug_phase_cycles(buf)
    v = list(buf)
    q = max(v[0], 1)
    print("%-40s queries %.3g  unseeded %.1f %%  saved %.1f %%  steps per seeded query %.2f, per unseeded search %.2f, wave-steps per 64 queries %.1f"
          % (what[:40], v[0], 100.0 * v[1] / q, 100.0 * v[2] / q, v[3] / max(v[0] - v[1] - v[2], 1), v[4] / max(v[1], 1), 64.0 * v[5] / q), flush=True)
    return e
bench.p2p_entry = entry
bench.secondary_logs(torch, capi)
