#!/usr/bin/env python3
"""Diagnostic (not product): build liblsa_core with -DLSA_PHASE_STAMPS into a scratch
path and print where icp_kernel's workgroups spend their cycles, per phase."""
import ctypes as C, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
out = os.path.join(ROOT, "gpurun_out", "liblsa_core_stamps.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
EXP = [a for a in sys.argv[1:] if a.startswith("-D")]
sys.argv = [a for a in sys.argv if not a.startswith("-D")]
subprocess.check_call(["hipcc"] + ge.HIP_FLAGS + ([] if "-DNOSTAMPS" in EXP else ["-DLSA_PHASE_STAMPS"]) + [e for e in EXP if e != "-DNOSTAMPS"] + ["-o", out,
                      os.path.join(ge.CSRC, "core.hip"), os.path.join(ge.CSRC, "carmen.cpp")])
from laser_slam_amd import capi, synth
capi.LIB_PATH = out
import torch
raw = np.concatenate([synth.make_scans(200, noise_seed=12345 + j)[0] for j in range(2)], 0)
p = capi.param_make(**synth.SYN1081)
ctx = capi.Context(p, 0)
S, n = raw.shape
d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda")
d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
ctx.preprocess_batch(S, d_r, d_bad, d_seg)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
idx = np.arange(B) % (S - 1)
idx = idx[(idx % 200) != 199]
B = len(idx)
ri = torch.from_numpy(idx.astype(np.int32)).cuda(); ai = ri + 1
pose = torch.zeros(B, 3, device="cuda")
its = torch.zeros(B, dtype=torch.int32, device="cuda")
ctx.match_batch(1, B, d_r, d_bad, d_seg, ri, ai, pose); ctx.sync()
buf = (C.c_ulonglong * 16)()
capi.lib().lsa_debug_phase_cycles(buf)
ctx.match_batch(1, B, d_r, d_bad, d_seg, ri, ai, pose, d_iters=its); ctx.sync()
capi.lib().lsa_debug_phase_cycles(buf)
mean_it = float(its.float().mean())
v = np.array(list(buf)[:8], np.float64)
names = ["project+scatter", "nn search", "compaction", "trim (bubble)", "owner+interp -> V", "sum means", "products+sum cov", "pose update"]
for nm, x in zip(names, v):
    print("%-22s %6.2f %%   %10.0f cycles/pair" % (nm, 100 * x / v.sum(), x / B))
print("total cycles/pair %.0f  mean iterations %.2f  cycles/iteration %.0f" % (v.sum() / B, mean_it, v.sum() / B / mean_it))
sub = np.array(list(buf)[8:16], np.float64) / B / mean_it
print("trim sub-phases (per iteration): first-k sort=%.0f | rank cand=%.0f bsearch=%.0f count g2+rank ent=%.0f count h/push+writes=%.0f emit=%.0f"
      % (sub[5], sub[6], sub[7], sub[0], sub[1], sub[2]))
print("per-iteration:", " ".join("%s=%.0f" % (nm.split()[0], x / B / mean_it) for nm, x in zip(names, v)))
st = list(buf)
steps = (st[12] >> 40) / B / mean_it
print("trim steps per iteration %.2f, candidates per step %.1f, entering per step %.1f" % (steps, st[11] / max(st[12] >> 40, 1), (st[12] & ((1 << 40) - 1)) / max(st[12] >> 40, 1)))
