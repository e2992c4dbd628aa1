#!/usr/bin/env python3
"""Generate the golden vectors in tests/golden/ from the REFERENCE built in place
(oracle/_ref, see oracle/Makefile).  Runs only in the build container, where
/root/reference exists; the outputs (data only) are committed.

Outputs
  <log>_head.log          first HEAD_SCANS laser records of each shipped CARMEN log
                          (input data of the reference's own drivers, data/*.log)
  zhpsm_<name>.npz        reader output, preprocess output, per-pair pm_psm / pm_icp /
                          pm_error_index2 / FMatch results, stage-level vectors
  pmap_<name>.npz         CPMap hit/sum/prob layers after inserting the first scans
  indoor_head.txt         first records of src/Main-Ctrl/ReadSLAMResult/indoor.txt
                          (the "LMS151" pose-less format) + reference result
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)

from oracle_lib import Ref, RefPMap, have_ref  # noqa: E402
from laser_slam_amd import synth  # noqa: E402

REF_DATA = "/root/reference/data"
HEAD_SCANS = 90
LOGS = [("fr079", "LMS511", "fr079.log"), ("intel", "LMS211", "intel-lab.log"), ("mit", "LMS511", "mit-cscail.log")]


def head_of_log(src, dst, tag, count):
    kept = 0
    with open(src) as f, open(dst, "w") as g:
        for line in f:
            if line.startswith(tag + " "):
                if kept >= count:
                    break
                kept += 1
                g.write(line)
            elif kept == 0 or not line.startswith(("VERTEX2", "EDGE2")):
                # keep a few non-laser lines so the "skip other tags" path is exercised
                if kept < 3:
                    g.write(line)
    return kept


def matcher_vectors(ref, raw, bad_raw, n_stage=12, seed=0):
    S = raw.shape[0]
    n = raw.shape[1]
    pre = [ref.preprocess(raw[k], bad_raw[k]) for k in range(S)]
    out = {
        "raw_r": raw, "raw_bad": bad_raw,
        "pre_r": np.stack([p[0] for p in pre]), "pre_bad": np.stack([p[1] for p in pre]),
        "pre_seg": np.stack([p[2] for p in pre]),
        "median_only": np.stack([ref.median(raw[k]) for k in range(min(S, 8))]),
    }
    B = S - 1
    for which, nm in ((0, "psm"), (1, "icp")):
        pose = np.zeros((B, 3), np.float32); ret = np.zeros(B, np.float32); st = np.zeros(B, np.int32)
        exy = np.full((B, 2), np.nan, np.float32); na = np.zeros(B, np.int32)
        for b in range(B):
            s, p, r = ref.match(which, pre[b], pre[b + 1])
            st[b], pose[b], ret[b] = s, p, r
            if s == 0:
                e, cnt = ref.error_index2(pre[b], pre[b + 1], (0, 0, 0), p)
                na[b] = cnt
                if cnt > 0:
                    exy[b] = e
        out[nm + "_pose"] = pose; out[nm + "_ret"] = ret; out[nm + "_status"] = st
        out[nm + "_exy"] = exy; out[nm + "_nassoc"] = na
    # non-zero initial poses (runlogImproved-style prior, ZHPolar_Match.cpp:785-789)
    rng = np.random.default_rng(seed)
    K = min(B, 16)
    init = np.zeros((K, 6), np.float32)
    init[:, 0:2] = rng.normal(0, 30, (K, 2)); init[:, 2] = rng.normal(0, 0.3, K)
    init[:, 3:5] = init[:, 0:2] + rng.normal(0, 8, (K, 2)); init[:, 5] = init[:, 2] + rng.normal(0, 0.05, K)
    out["init_pose"] = init
    for which, nm in ((0, "psm"), (1, "icp")):
        pose = np.zeros((K, 3), np.float32); st = np.zeros(K, np.int32)
        for b in range(K):
            st[b], pose[b], _ = ref.match(which, pre[b], pre[b + 1], init[b, :3], init[b, 3:])
        out[nm + "_init_pose_out"] = pose; out[nm + "_init_status"] = st
    # FMatch chain: scan k is preprocessed once as "act" and reused as "ref" 
    fm_pose = np.zeros((B, 3), np.float32); fm_pair = np.zeros((B, 2), np.float32)
    cur = [raw[0].copy(), bad_raw[0].copy(), np.zeros(n, np.int32)]
    for b in range(B):
        act = [raw[b + 1].copy(), bad_raw[b + 1].copy(), np.zeros(n, np.int32)]
        p, pair, r2, a2 = ref.fmatch(cur, act, b == 0)
        fm_pose[b], fm_pair[b] = p, pair
        cur = a2
    out["fmatch_pose"] = fm_pose; out["fmatch_pair"] = fm_pair
    # stage-level vectors
    idx = np.linspace(1, S - 1, n_stage).astype(int)
    sp_pose = np.zeros((n_stage, 3), np.float32)
    sp_pose[:, :2] = rng.normal(0, 12, (n_stage, 2)); sp_pose[:, 2] = rng.normal(0, 0.15, n_stage)
    sp_r = np.zeros((n_stage, n), np.float32); sp_bad = np.zeros((n_stage, n), np.int32)
    osr = np.zeros((n_stage, 2), np.float32); tre = np.zeros((n_stage, 4), np.float32)
    for t, k in enumerate(idx):
        sp_r[t], sp_bad[t] = ref.scan_project(*pre[k], sp_pose[t])
        s, dth = ref.orientation_search(pre[k - 1][0], pre[k - 1][1], sp_r[t], sp_bad[t])
        osr[t] = (s, dth)
        tre[t] = ref.translation_estimation(pre[k - 1][0], pre[k - 1][1], sp_r[t], sp_bad[t])
    out.update(stage_idx=idx.astype(np.int32), stage_pose=sp_pose, stage_new_r=sp_r, stage_new_bad=sp_bad,
               stage_orient=osr, stage_trans=tre)
    return out


def main():
    if not have_ref():
        raise SystemExit("oracle/_ref missing: run `make -C oracle ref` in the build container")
    np.set_printoptions(suppress=True)
    for name, laser, fn in LOGS:
        dst = os.path.join(HERE, name + "_head.log")
        kept = head_of_log(os.path.join(REF_DATA, fn), dst, "ROBOTLASER1", HEAD_SCANS)
        ref = Ref(laser)
        r, x, y, bad, pose = ref.read_carmen(dst)
        assert r.shape[0] == kept
        vec = matcher_vectors(ref, r, bad)
        vec.update(reader_x=x, reader_y=y, reader_pose=pose,
                   param=np.array(list(ref.param_dict().values()), np.float64),
                   param_keys=np.array(list(ref.param_dict().keys())))
        fi, si, co = ref.tables()
        vec.update(tab_fi=fi, tab_si=si, tab_co=co)
        np.savez_compressed(os.path.join(HERE, "zhpsm_%s.npz" % name), **vec)
        print(name, "scans", kept, "psm fail", int((vec["psm_status"] != 0).sum()),
              "icp fail", int((vec["icp_status"] != 0).sum()))

        # CPMap: insert the first scans at the log's laser poses (global metres)
        if name in ("mit", "fr079"):
            res = 0.05 if name == "mit" else 0.02
            G = 512
            p0 = pose[0]
            pm = RefPMap(G, G, G // 2, G // 2, float(p0[0]), float(p0[1]), res)
            n_ins = 30
            gxs, gys, rob = [], [], []
            for k in range(n_ins):
                rr = r[k] / 100.0
                ok = rr <= ref.param_dict()["max_range"] / 100.0
                th = pose[k, 2]
                gx = (pose[k, 0] + rr * np.cos(th + fi)).astype(np.float32)
                gy = (pose[k, 1] + rr * np.sin(th + fi)).astype(np.float32)
                gx, gy = gx[ok], gy[ok]
                pm.update(gx, gy, float(pose[k, 0]), float(pose[k, 1]))
                pad = np.zeros(r.shape[1], np.float32)
                gxp, gyp = pad.copy(), pad.copy()
                gxp[:len(gx)] = gx; gyp[:len(gy)] = gy
                gxs.append(gxp); gys.append(gyp); rob.append((pose[k, 0], pose[k, 1], len(gx)))
            pm.compute()
            hit, sm, prob = pm.layers()
            cx, cy = pm.cloud()
            np.savez_compressed(os.path.join(HERE, "pmap_%s.npz" % name), gx=np.stack(gxs), gy=np.stack(gys),
                                robot=np.array(rob, np.float32), hit=hit.astype(np.uint16), sum=sm.astype(np.uint16),
                                prob=prob, cloud_x=cx, cloud_y=cy,
                                geom=np.array([G, G, G // 2, G // 2, p0[0], p0[1], res], np.float64))
            print(name, "pmap sum", sm.sum(), "hit", hit.sum(), "cloud", len(cx))

    # synthetic 1081-beam set (parameters injected identically everywhere)
    raw, poses = synth.make_scans(40)
    ref = Ref(**synth.SYN1081)
    vec = matcher_vectors(ref, raw, np.zeros(raw.shape, np.int32), seed=1)
    vec.update(synth_pose=poses, param=np.array(list(ref.param_dict().values()), np.float64),
               param_keys=np.array(list(ref.param_dict().keys())))
    fi, si, co = ref.tables()
    vec.update(tab_fi=fi, tab_si=si, tab_co=co)
    np.savez_compressed(os.path.join(HERE, "zhpsm_syn1081.npz"), **vec)
    print("syn1081 psm fail", int((vec["psm_status"] != 0).sum()), "icp fail", int((vec["icp_status"] != 0).sum()))

    # the in-house pose-less "LMS151" format (readCarmon with an "LMS" tag, :204-211)
    src = "/root/reference/src/Main-Ctrl/ReadSLAMResult/indoor.txt"
    dst = os.path.join(HERE, "indoor_head.txt")
    with open(src) as f, open(dst, "w") as g:
        for k, line in enumerate(f):
            if k >= 24:
                break
            g.write(line.replace("\t", " "))
    ref = Ref("LMS151")
    r, x, y, bad, pose = ref.read_carmen(dst, "LMS151")
    vec = matcher_vectors(ref, r, bad, n_stage=4, seed=2)
    vec.update(param=np.array(list(ref.param_dict().values()), np.float64),
               param_keys=np.array(list(ref.param_dict().keys())))
    np.savez_compressed(os.path.join(HERE, "zhpsm_indoor.npz"), **vec)
    print("indoor scans", r.shape[0])


if __name__ == "__main__":
    main()
