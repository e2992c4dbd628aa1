#!/usr/bin/env python3
"""One-off soak (diagnostic, uses the oracle as the checker): many random pairs with random priors, both
matchers, device against CPU oracle, counting status / bit mismatches."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from laser_slam_amd import capi, synth
from oracle_lib import SYN1081, Oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 1081          # 1081 (the benchmark sensor) or one of the reference's sizes
if NB != 1081:
    _w = {181: 20, 361: 40, 541: 50}[NB]
    SYN1081 = dict(points=NB, fov_deg=270, max_range=3000, min_valid=NB // 4, window=_w, fi_min_deg=-135, fi_max_deg=135)
    synth.SYN1081 = SYN1081
raw = np.concatenate([synth.make_scans(60, noise_seed=synth.WORLD_SEED + 500 + j, **({} if NB == 1081 else dict(n_beams=NB)))[0] for j in range(6)], 0)
o = Oracle(**SYN1081)
pre = [o.preprocess(raw[k]) for k in range(len(raw))]
r = np.stack([q[0] for q in pre]); bad = np.stack([q[1] for q in pre]); seg = np.stack([q[2] for q in pre])
base = rng.integers(0, 6, B) * 60
a = rng.integers(0, 60, B); b = np.clip(a + rng.integers(-12, 13, B), 0, 59)
ri, ai = (base + a).astype(np.int32), (base + b).astype(np.int32)
init = np.zeros((B, 6), np.float32)
init[:, 0:2] = rng.uniform(-300, 300, (B, 2)); init[:, 2] = rng.uniform(-3, 3, B)
init[:, 3:5] = init[:, 0:2] + rng.uniform(-30, 30, (B, 2)); init[:, 5] = init[:, 2] + rng.uniform(-0.08, 0.08, B)
init[: B // 4] = 0
ctx = capi.Context(capi.param_make(**synth.SYN1081))
d = lambda x, t: torch.from_numpy(np.ascontiguousarray(x).astype(t)).cuda()
d_r, d_bad, d_seg = d(r, np.float32), d(bad, np.uint8), d(seg, np.int16)
d_ri, d_ai, d_init = d(ri, np.int32), d(ai, np.int32), d(init, np.float32)
for which in (0, 1):
    pose = torch.zeros(B, 3, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    ctx.match_batch(which, B, d_r, d_bad, d_seg, d_ri, d_ai, pose, d_init, d_status=st); ctx.sync()
    gp, gs = pose.cpu().numpy(), st.cpu().numpy()
    t0 = time.time()
    def work(rng_):
        oo = Oracle(**SYN1081)
        badk = []
        for k in rng_:
            s_, p_, _, _ = oo.match(which, pre[ri[k]], pre[ai[k]], init[k, :3], init[k, 3:])
            if s_ != gs[k] or (s_ == 0 and p_.tobytes() != gp[k].tobytes()):
                badk.append(k)
        return badk
    chunks = np.array_split(np.arange(B), 16)
    with ThreadPoolExecutor(16) as ex:
        bads = sum(ex.map(work, chunks), [])
    print("matcher %s: %d pairs, %d failures of the matcher, %d mismatches (%.0f s)" % ("psm" if which == 0 else "icp", B, int((gs != 0).sum()), len(bads), time.time() - t0), flush=True)
    for k in bads[:5]:
        print("   pair", k, ri[k], ai[k], init[k], gs[k], gp[k])
