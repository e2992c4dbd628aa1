#!/usr/bin/env python3
"""Diagnostic: pm_icp of a build (argv[1] = library, default the in-tree one) on random synthetic pairs with random
priors against the CPU oracle, on padded device arrays so that a wild beam index cannot leave the allocation; with a
-DLSA_CHECK_INDEX build the wild-index counters are printed too.  Prints the number of mismatches."""
import ctypes as C, os, sys
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from laser_slam_amd import capi, synth
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
import torch
from oracle_lib import SYN1081, Oracle
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
rng = np.random.default_rng(3)
raw = np.concatenate([synth.make_scans(60, noise_seed=synth.WORLD_SEED + 700 + j)[0] for j in range(3)], 0)
o = Oracle(**SYN1081)
pre = [o.preprocess(raw[k]) for k in range(len(raw))]
S, n = raw.shape
PAD = 200                                                   # rows behind the scan set
r = np.zeros((S + PAD, n), np.float32); bad = np.zeros((S + PAD, n), np.uint8); seg = np.zeros((S + PAD, n), np.int16)
r[:S] = np.stack([q[0] for q in pre]); bad[:S] = np.stack([q[1] for q in pre]); seg[:S] = np.stack([q[2] for q in pre])
base = rng.integers(0, 3, B) * 60
a = rng.integers(0, 60, B); b = np.clip(a + rng.integers(-10, 11, B), 0, 59)
ri, ai = (base + a).astype(np.int32), (base + b).astype(np.int32)
init = np.zeros((B, 6), np.float32)
init[:, 0:2] = rng.uniform(-200, 200, (B, 2)); init[:, 2] = rng.uniform(-3, 3, B)
init[:, 3:5] = init[:, 0:2] + rng.uniform(-25, 25, (B, 2)); init[:, 5] = init[:, 2] + rng.uniform(-0.06, 0.06, B)
init[: B // 3] = 0
ctx = capi.Context(capi.param_make(**synth.SYN1081))
d = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
d_r, d_bad, d_seg, d_ri, d_ai, d_init = d(r), d(bad), d(seg), d(ri), d(ai), d(init)
pose = torch.zeros(B, 3, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda"); ret = torch.zeros(B, device="cuda")
exy = torch.zeros(B, 2, device="cuda"); na = torch.zeros(B, dtype=torch.int32, device="cuda"); it = torch.zeros(B, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
ctx.match_batch(1, B, d_r, d_bad, d_seg, d_ri, d_ai, pose, d_init, ret, exy, na, st, it); ctx.sync()
gp, gs = pose.cpu().numpy(), st.cpu().numpy()
if hasattr(capi.lib(), "lsa_debug_bad_index"):
    buf = (C.c_uint * 4)(); capi.lib().lsa_debug_bad_index(buf); print("wild indices (ref, cur):", list(buf)[:2])


def work(ks):
    oo = Oracle(**SYN1081)
    return [k for k in ks if (lambda s_, p_: s_ != gs[k] or (s_ == 0 and p_.tobytes() != gp[k].tobytes()))(*oo.match(1, pre[ri[k]], pre[ai[k]], init[k, :3], init[k, 3:])[:2])]


with ThreadPoolExecutor(16) as ex:
    bads = sum(ex.map(work, np.array_split(np.arange(B), 16)), [])
print("%s: %d pairs, %d matcher failures, mean iterations %.2f, %d MISMATCHES" % (os.path.basename(capi.LIB_PATH), B, int((gs != 0).sum()), float(it.float().mean()), len(bads)))
for k in bads[:4]:
    print("   pair", k, ri[k], ai[k], gs[k], gp[k])
