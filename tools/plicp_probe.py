#!/usr/bin/env python3
"""Diagnostic: one batch of CCanonicalMatcher's point-to-line ICP (for rocprofv3 --pmc runs)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from laser_slam_amd import capi, synth
if len(sys.argv) > 1 and sys.argv[1] != "-":
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
if len(sys.argv) > 3:                                       # a log's head instead of the synthetic set: fr079 | intel | mit | indoor
    preset, fname, tag = {"fr079": ("LMS511", "fr079_head.log", "ROBOTLASER1"), "intel": ("LMS211", "intel_head.log", "ROBOTLASER1"),
                          "mit": ("LMS511", "mit_head.log", "ROBOTLASER1"), "indoor": ("LMS151", "indoor_head.txt", "LMS151")}[sys.argv[3]]
    p = capi.param_preset(preset)
    raw = capi.carmen_read(p, os.path.join(ROOT, "tests", "golden", fname), tag)[0]
    ctx = capi.Context(p, 0)
    idx = (np.arange(B) % (len(raw) - 1)).astype(np.int32)
else:
    raw = np.concatenate([synth.make_scans(200, noise_seed=12345 + j)[0] for j in range(2)], 0)
    p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
    idx = np.arange(B) % (len(raw) - 1); idx = np.where(idx % 200 == 199, idx - 1, idx).astype(np.int32)
d_r = torch.from_numpy(raw).cuda(); ri = torch.from_numpy(idx).cuda(); ai = ri + 1
out = torch.zeros(B, 8, dtype=torch.float64, device="cuda"); torch.cuda.synchronize()
capi.plicp_batch(ctx, B, d_r, ri, ai, out); ctx.sync()
reps = 3 if B <= 32768 else 1
t0 = time.perf_counter()
for _ in range(reps): capi.plicp_batch(ctx, B, d_r, ri, ai, out)
ctx.sync(); dt = (time.perf_counter() - t0) / reps
print("plicp B=%d: %.0f matches/s, mean iterations %.2f, valid %.3f" % (B, B / dt, float(out[:, 5].mean()), float(out[:, 3].mean())))
if hasattr(capi.lib(), "lsa_debug_plicp_cycles"):
    import ctypes as C
    buf = (C.c_ulonglong * 8)()
    capi.lib().lsa_debug_plicp_cycles(buf)
    v = np.array(list(buf), np.float64) / ((1 + reps) * B) / float(out[:, 5].mean())
    print("cycles per iteration: nn=%.0f doubles=%.0f trim=%.0f sums=%.0f solve=%.0f hash/stop=%.0f | inside nn: plan+seeds %.0f, window scan %.0f (the rest of nn: second point, records)" % tuple(v[:8]))
    raw6, raw7 = int(buf[6]), int(buf[7])
    if os.environ.get("LSA_PLICP_COUNTS"):
        its = (1 + reps) * B * float(out[:, 5].mean())
        print("per iteration: slow bounded %.1f, slow unbounded %.1f, candidates %.0f, points with > 64 candidates %.1f"
              % ((raw6 & 0xfffff) / its, (raw6 >> 20) / its, (raw7 & 0xffffffff) / its, (raw7 >> 32) / its))
