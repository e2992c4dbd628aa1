#!/usr/bin/env python3
"""Diagnostic: candidate-list statistics of icp_kernel's NN search (build_ab/nnstats.so, -DLSA_NN_STATS)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import capi, synth
capi.LIB_PATH = os.path.join(ROOT, "build_ab", "nnstats.so")
import torch
raw = np.concatenate([synth.make_scans(200, noise_seed=12345 + j)[0] for j in range(2)], 0)
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
S, n = raw.shape
d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
torch.cuda.synchronize(); ctx.preprocess_batch(S, d_r, d_bad, d_seg)
B = 2048
idx = np.arange(B) % (S - 1); idx = np.where(idx % 200 == 199, idx - 1, idx)
ri = torch.from_numpy(idx.astype(np.int32)).cuda(); ai = ri + 1
pose = torch.zeros(B, 3, device="cuda"); torch.cuda.synchronize()
buf = (C.c_ulonglong * 24)()
capi.lib().lsa_debug_nn_stats(buf)
ctx.match_batch(1, B, d_r, d_bad, d_seg, ri, ai, pose); ctx.sync()
capi.lib().lsa_debug_nn_stats(buf)
v = [float(x) for x in buf]
print("blocks %d | live lanes/block %.1f | candidates per live lane %.2f | blocks entering the general loop %.1f %% | "
      "max list of non-flat lanes (mean over all blocks) %.1f | non-flat lanes/block %.2f, their mean list %.1f | flat max len %.2f"
      % (v[0], v[4] / v[0], v[1] / max(v[4], 1), 100 * v[2] / v[0], v[3] / v[0], v[5] / v[0], v[6] / max(v[5], 1), v[7] / v[0]))
print("live lanes by list length (<=8, <=16, <=32, <=64, <=128, >128): " + " ".join("%.2f%%" % (100 * v[8 + k] / v[4]) for k in range(6)), "| unbounded lanes %.3f%%" % (100 * v[14] / v[4]))
print("blocks by longest non-flat list: " + " ".join("%.1f%%" % (100 * v[16 + k] / v[0]) for k in range(6)))
print("with the final best distance as the bound: candidates per live lane %.2f, mean over blocks of the longest list >8: %.1f" % (v[15] / v[4], v[22] / v[0]))
