#!/usr/bin/env python3
"""Diagnostic: one ICP (or PSM) batch launch through a prebuilt variant of liblsa_core.so, for
rocprofv3 --pmc runs (`rocprofv3 --pmc SQ_INSTS_VALU ... -- python3 tools/pmc_variant.py --lib X`)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--lib", default=None)
ap.add_argument("--which", default="icp")
ap.add_argument("--pairs", type=int, default=8192)
a = ap.parse_args()
from laser_slam_amd import capi, synth
if a.lib:
    capi.LIB_PATH = os.path.abspath(a.lib)
import torch
raw = np.concatenate([synth.make_scans(200, noise_seed=12345 + j)[0] for j in range(4)], 0)
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
S, n = raw.shape
d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
torch.cuda.synchronize(); ctx.preprocess_batch(S, d_r, d_bad, d_seg)
B = a.pairs
idx = np.arange(B) % (S - 1); idx = np.where(idx % 200 == 199, idx - 1, idx)
ri = torch.from_numpy(idx.astype(np.int32)).cuda(); ai = ri + 1
pose = torch.zeros(B, 3, device="cuda"); it = torch.zeros(B, dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
ctx.match_batch(capi.MATCH_ICP if a.which == "icp" else capi.MATCH_PSM, B, d_r, d_bad, d_seg, ri, ai, pose, d_iters=it); ctx.sync()
print("pairs", B, "mean iterations", float(it.float().mean()))
