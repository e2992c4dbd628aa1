#!/usr/bin/env python3
"""Diagnostic: scan simulator / beam model rate, and bit comparison of two builds (argv: libA [libB])."""
import os, sys, time, subprocess, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2:          # compare two builds in separate processes
    outs = [subprocess.run([sys.executable, __file__, lib], capture_output=True, text=True).stdout for lib in sys.argv[1:3]]
    a, b = (json.loads(o.strip().splitlines()[-1]) for o in outs)
    print(outs[0].strip().splitlines()[0]); print(outs[1].strip().splitlines()[0])
    print("identical ranges:", a["r"] == b["r"], " identical flags:", a["bad"] == b["bad"], " identical scores:", a["score"] == b["score"])
    sys.exit(0)
from laser_slam_amd import capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
import torch, hashlib
rng = np.random.default_rng(0)
sx, sy, res = 800, 600, 0.05
prob = np.full((sx, sy), 0.1, np.float32)
prob[:, :3] = 0.9; prob[:, -3:] = 0.9; prob[:3, :] = 0.9; prob[-3:, :] = 0.9
for _ in range(25):
    a, b = rng.integers(20, sx - 60), rng.integers(20, sy - 60)
    prob[a:a + rng.integers(6, 40), b:b + rng.integers(6, 40)] = 0.95
prob[300:340, 200:260] = 0.5
ctx = capi.Context(capi.param_preset("LMS151"))
P, n = 20000, 541
poses = np.stack([rng.uniform(2, sx * res - 2, P), rng.uniform(2, sy * res - 2, P), rng.uniform(-3.14, 3.14, P)], 1).astype(np.float32)
d_prob = torch.from_numpy(prob).cuda(); d_poses = torch.from_numpy(poses).cuda()
r = torch.zeros(P, n, device="cuda"); bad = torch.zeros(P, n, dtype=torch.uint8, device="cuda")
fld = capi.Field(ctx, sx, sy, 0, 0, 0.0, 0.0, res)
torch.cuda.synchronize()
capi.simulate_scans(fld, d_prob, d_poses, P, n, r, bad); ctx.sync()
t0 = time.perf_counter()
for _ in range(3): capi.simulate_scans(fld, d_prob, d_poses, P, n, r, bad)
ctx.sync(); dt = (time.perf_counter() - t0) / 3
obs_r = r[0].clone(); obs_bad = bad[0].clone()
score = torch.zeros(P, device="cuda")
capi.beam_model_scores(fld, d_prob, d_poses, P, obs_r, obs_bad, n, score); ctx.sync()
t0 = time.perf_counter()
for _ in range(3): capi.beam_model_scores(fld, d_prob, d_poses, P, obs_r, obs_bad, n, score)
ctx.sync(); dt2 = (time.perf_counter() - t0) / 3
print("%s: simulator %.0f poses/s (%.0f M rays/s), mean range %.2f m; beam model %.0f poses/s" % (os.path.basename(sys.argv[1]), P / dt, P * n / dt / 1e6, float(r.mean()), P / dt2))
h = lambda t: hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()
print(json.dumps(dict(r=h(r), bad=h(bad), score=h(score))))
