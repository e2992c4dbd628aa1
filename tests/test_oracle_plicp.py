"""CPU: the point-to-line ICP oracle (oracle/plicp_oracle.c, CCanonicalMatcher's options) against first
principles - CSM is absent, so the restatement is pinned to motions that are known by construction."""
import numpy as np

import oracle_lib as ol
from laser_slam_amd import synth


def test_plicp_recovers_synthetic_motion_and_agrees_with_pm_icp():
    raw, poses = synth.make_scans(6)
    o = ol.Oracle(**ol.SYN1081)
    P = o.param_dict()
    for k in range(1, 6):
        out = ol.plicp(raw[k - 1], raw[k], P["fi_min"], P["dfi"], P["min_range"], P["max_range"])
        x0, y0, t0 = poses[k - 1]; x1, y1, t1 = poses[k]
        c, s = np.cos(-t0), np.sin(-t0)
        true = np.array([c * (x1 - x0) - s * (y1 - y0), s * (x1 - x0) + c * (y1 - y0), t1 - t0])
        assert out[3] == 1.0 and 1 <= out[5] <= 70 and out[6] > 0.5 * 1081
        assert np.abs(out[:2] - true[:2]).max() < 0.01 and abs(out[2] - true[2]) < 0.003
        st, pose, _, _ = o.match(1, o.preprocess(raw[k - 1]), o.preprocess(raw[k]))
        assert st == 0 and np.abs(out[:2] - pose[:2] / 100.0).max() < 0.01 and abs(out[2] - pose[2]) < 0.003


def test_plicp_identity_and_failure():
    raw, _ = synth.make_scans(2)
    o = ol.Oracle(**ol.SYN1081)
    P = o.param_dict()
    same = ol.plicp(raw[0], raw[0], P["fi_min"], P["dfi"], P["min_range"], P["max_range"])
    assert same[3] == 1.0 and np.abs(same[:3]).max() < 1e-6 and same[4] < 1e-6
    blind = np.full(1081, 3001.0, np.float32)                       # nothing in range: fewer than 5 % correspondences
    bad = ol.plicp(raw[0], blind, P["fi_min"], P["dfi"], P["min_range"], P["max_range"])
    assert bad[3] == 0.0
