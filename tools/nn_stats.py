#!/usr/bin/env python3
"""Diagnostic (GPU box): how often pm_icp's nearest-neighbour search leaves its straight-line path (a -DLSA_PHASE_STAMPS
-DLSA_NN_STATS build): wave-trips that run the general path, lanes on it, its steps by lanes and by waves."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
out = os.path.join(ROOT, "gpurun_out", "liblsa_core_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call(["hipcc"] + ge.HIP_FLAGS + ["-DLSA_PHASE_STAMPS", "-DLSA_NN_STATS", "-o", out, os.path.join(ge.CSRC, "core.hip"), os.path.join(ge.CSRC, "carmen.cpp")])
from laser_slam_amd import capi, synth
capi.LIB_PATH = out
import torch
raw = np.concatenate([synth.make_scans(200, noise_seed=12345 + j)[0] for j in range(2)], 0)
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
S, n = raw.shape
d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
ctx.preprocess_batch(S, d_r, d_bad, d_seg)
B = 4096
idx = np.arange(B) % (S - 1); idx = idx[(idx % 200) != 199]; B = len(idx)
ri = torch.from_numpy(idx.astype(np.int32)).cuda(); ai = ri + 1
pose = torch.zeros(B, 3, device="cuda"); its = torch.zeros(B, dtype=torch.int32, device="cuda")
buf = (C.c_ulonglong * 16)()
ctx.match_batch(1, B, d_r, d_bad, d_seg, ri, ai, pose); ctx.sync()
capi.lib().lsa_debug_phase_cycles(buf)
ctx.match_batch(1, B, d_r, d_bad, d_seg, ri, ai, pose, d_iters=its); ctx.sync()
capi.lib().lsa_debug_phase_cycles(buf)
v = list(buf); iters = float(its.sum())
trips = iters * 17                       # 64-beam chunks per iteration at 1081 beams
gen = v[12]
print("wave-trips %.3g, on the general path %.1f %%; lanes on it per such trip %.1f; its steps: %.1f per lane on it, %.1f per wave-trip on it"
      % (trips, 100.0 * gen / trips, v[15] / max(gen, 1), v[13] / max(v[15], 1), v[14] / max(gen, 1)))
