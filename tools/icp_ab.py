#!/usr/bin/env python3
"""Diagnostic (GPU box): pm_icp (LSA_AB_MATCHER=0: pm_psm) rate of several builds on the synthetic 1081-beam set, one process per build, interleaved.
usage: icp_ab.py [--pairs N] [--rounds R] build_ab/a.so build_ab/b.so ...  (a build named `-` is the product's library)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--worker" in sys.argv:
    import numpy as np, torch
    import bench
    from laser_slam_amd import capi, synth
    lib, B = sys.argv[sys.argv.index("--worker") + 1], int(sys.argv[sys.argv.index("--worker") + 2])
    if lib != "-": capi.LIB_PATH = os.path.abspath(lib)
    raw = bench.build_scan_set(); S, n = raw.shape
    p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
    d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
    ctx.preprocess_batch(S, d_r, d_bad, d_seg)
    base_pairs = np.array([(q * bench.N_BASE + k - 1, q * bench.N_BASE + k) for q in range(bench.SEEDS) for k in range(1, bench.N_BASE)], np.int32)
    sel = np.arange(B) % len(base_pairs)
    ri = torch.from_numpy(np.ascontiguousarray(base_pairs[sel, 0])).cuda(); ai = torch.from_numpy(np.ascontiguousarray(base_pairs[sel, 1])).cuda()
    pose = torch.zeros(B, 3, device="cuda")
    its = torch.zeros(B, dtype=torch.int32, device="cuda")
    ms = bench.timed(torch, lambda: ctx.match_batch(int(os.environ.get("LSA_AB_MATCHER", "1")), B, d_r, d_bad, d_seg, ri, ai, pose, d_iters=its), reps=3)
    print("%.4f %.3f" % (B / ms / 1e3, float(its.float().mean())), flush=True)
    sys.exit(0)
args = sys.argv[1:]
B = int(args[args.index("--pairs") + 1]) if "--pairs" in args else 262144
R = int(args[args.index("--rounds") + 1]) if "--rounds" in args else 2
libs = [a for i, a in enumerate(args) if not a.startswith("--") and (i == 0 or args[i - 1] not in ("--pairs", "--rounds"))]
res = {l: [] for l in libs}
for r in range(R):
    for l in libs:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--worker", l, str(B)], capture_output=True, text=True, timeout=300)
        if out.returncode: print(l, "failed:", out.stderr[-400:]); continue
        a, b = out.stdout.strip().split("\n")[-1].split()
        res[l].append((float(a), float(b)))
for l in libs: print("%-28s %s M matches/s  (%.2f iterations, %.2f M iterations/s)" % (l, " ".join("%.4f" % x[0] for x in res[l]), res[l][0][1], max(x[0] for x in res[l]) * res[l][0][1]), flush=True)
