#!/usr/bin/env python3
"""Diagnostic: build liblsa_core with extra -D flags into a scratch path and time icp/psm batches."""
import os, subprocess, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
flags = [a for a in sys.argv[1:] if a.startswith("-D")]
libs = [a[6:] for a in sys.argv[1:] if a.startswith("--lib=")]      # prebuilt variant (tools/build_variants.py)
if libs:
    out = os.path.abspath(libs[0]); flags = [os.path.basename(out)]
else:
    out = os.path.join(ROOT, "gpurun_out", "liblsa_core_ab.so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["hipcc"] + ge.HIP_FLAGS + flags + ["-o", out, os.path.join(ge.CSRC, "core.hip"), os.path.join(ge.CSRC, "carmen.cpp")])
from laser_slam_amd import capi, synth
capi.LIB_PATH = out
import torch
raw = np.concatenate([synth.make_scans(200, noise_seed=12345 + j)[0] for j in range(4)], 0)
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
S, n = raw.shape
d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
torch.cuda.synchronize(); ctx.preprocess_batch(S, d_r, d_bad, d_seg)
B = 32768
idx = np.arange(B) % (S - 1); idx = np.where(idx % 200 == 199, idx - 1, idx)
ri = torch.from_numpy(idx.astype(np.int32)).cuda(); ai = ri + 1
pose = torch.zeros(B, 3, device="cuda"); torch.cuda.synchronize()
for which, nm in ((1, "icp"), (0, "psm")):
    ctx.match_batch(which, B, d_r, d_bad, d_seg, ri, ai, pose); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(3): ctx.match_batch(which, B, d_r, d_bad, d_seg, ri, ai, pose)
    ctx.sync(); dt = (time.perf_counter() - t0) / 3
    print("%s %s: %.0f matches/s" % (" ".join(flags) or "base", nm, B / dt))
