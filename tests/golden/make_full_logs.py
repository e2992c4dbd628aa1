#!/usr/bin/env python3
"""Whole-log golden vectors from the REFERENCE built in place (oracle/_ref; build container only).

For every scan pair (k-1, k) of the three CARMEN logs the reference ships (data/fr079.log, intel-lab.log,
mit-cscail.log - BASELINE.json's configs 1, 2 and 5) this stores what the reference's own code returns:
pm_psm and pm_icp from a zero prior (pose, status), and the FMatch chain the front-ends run (pose + pair), with
the raw ranges and flags its CARMEN reader produced.  For mit-cscail additionally the CPMap (4096 x 4096 @ 5 cm) that
the reference's CPMap builds from the FMatch chain, as SHA-256 digests of its layers (config 5 at full length).

Outputs: tests/golden/full_<name>.npz - data only (ranges as uint16 where the log's 1 cm quantisation allows it).
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from helpers import chain_poses, scan_endpoints  # noqa: E402
from oracle_lib import Ref, RefPMap, have_ref  # noqa: E402

REF_DATA = "/root/reference/data"
LOGS = [("fr079", "LMS511", "fr079.log"), ("intel", "LMS211", "intel-lab.log"), ("mit", "LMS511", "mit-cscail.log")]
GRID = dict(size=4096, res=0.05)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    if not have_ref():
        raise SystemExit("oracle/_ref missing: run `make -C oracle ref` in the build container")
    for name, laser, fn in LOGS:
        t0 = time.time()
        ref = Ref(laser)
        r, x, y, bad, pose = ref.read_carmen(os.path.join(REF_DATA, fn))
        S, n = r.shape
        pre = [ref.preprocess(r[k], bad[k]) for k in range(S)]
        B = S - 1
        out = {"raw_bad_any": np.array([int(bad.any())]), "log_pose": pose,
               "param": np.array(list(ref.param_dict().values()), np.float64), "param_keys": np.array(list(ref.param_dict().keys()))}
        # ranges: the logs are quantised to 1 cm, so cm values are whole numbers below 65536 -> uint16 (checked)
        if np.array_equal(r, np.round(r)) and r.max() < 65536 and r.min() >= 0:
            out["raw_r_u16"] = r.astype(np.uint16)
        else:
            out["raw_r"] = r
        if bad.any():
            out["raw_bad"] = bad.astype(np.uint8)
        for which, nm in ((0, "psm"), (1, "icp")):
            p = np.zeros((B, 3), np.float32); st = np.zeros(B, np.int32)
            for b in range(B):
                s, q, _ = ref.match(which, pre[b], pre[b + 1])
                st[b], p[b] = s, q
            out[nm + "_pose"] = p; out[nm + "_status"] = st.astype(np.uint8)
        fm_pose = np.zeros((B, 3), np.float32); fm_pair = np.zeros((B, 2), np.float32)
        cur = [r[0].copy(), bad[0].copy(), np.zeros(n, np.int32)]
        for b in range(B):
            act = [r[b + 1].copy(), bad[b + 1].copy(), np.zeros(n, np.int32)]
            q, pair, _, a2 = ref.fmatch(cur, act, b == 0)
            fm_pose[b], fm_pair[b] = q, pair
            cur = a2
        out["fmatch_pose"] = fm_pose; out["fmatch_pair"] = fm_pair
        if name == "mit":
            failed = (fm_pair[:, 0] < 0) & (fm_pair[:, 1] < 0)
            gp = chain_poses(fm_pose, failed)
            fi = ref.tables()[0]
            G = GRID["size"]
            pm = RefPMap(G, G, G // 2, G // 2, 0.0, 0.0, GRID["res"])
            nv = np.zeros(S, np.int32)
            for k in range(S):
                gx, gy, cnt = scan_endpoints(pre[k][0], pre[k][1], fi, gp[k], ref.param_dict()["max_range"])
                nv[k] = cnt
                pm.update(gx[:cnt], gy[:cnt], float(np.float32(gp[k, 0])), float(np.float32(gp[k, 1])))
            pm.compute()
            hit, sm, prob = pm.layers()
            out.update(chain_pose=gp, map_n_valid=nv, map_hit_total=np.array([hit.astype(np.int64).sum()]),
                       map_sum_total=np.array([sm.astype(np.int64).sum()]),
                       map_sha=np.array([sha(hit.astype(np.int32)), sha(sm.astype(np.int32)), sha(prob.astype(np.float32))]),
                       map_geom=np.array([G, G, G // 2, G // 2, 0.0, 0.0, GRID["res"]], np.float64))
            print(name, "map hit", int(out["map_hit_total"][0]), "sum", int(out["map_sum_total"][0]))
        path = os.path.join(HERE, "full_%s.npz" % name)
        np.savez_compressed(path, **out)
        print(name, "pairs", B, "psm fail", int((out["psm_status"] != 0).sum()), "icp fail", int((out["icp_status"] != 0).sum()),
              "fmatch fail", int(((fm_pair[:, 0] < 0) & (fm_pair[:, 1] < 0)).sum()), "%.0f KB" % (os.path.getsize(path) / 1024),
              "%.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
