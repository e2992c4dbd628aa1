"""CPU: the oracle restatement against the committed golden vectors (generated from
the reference built in place by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, SETS, golden_param, load, same_bits
from oracle_lib import SYN1081, Oracle, OraclePMap, grid_line


def make_oracle(name, g):
    laser = SETS[name][0]
    o = Oracle(laser) if laser else Oracle(**SYN1081)
    gp = golden_param(g)
    o.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
    return o


@pytest.mark.parametrize("name", list(SETS))
def test_param_and_tables(name):
    g = load(name)
    o = make_oracle(name, g)
    gp = golden_param(g)
    od = o.param_dict()
    for k in ("points", "fov_deg", "max_range", "min_range", "min_valid", "window"):
        assert od[k] == int(gp[k]), k
    assert np.float32(od["dfi"]) == np.float32(gp["dfi"])
    if "tab_fi" in g:
        fi, si, co = o.tables()
        assert same_bits(fi, g["tab_fi"]) and same_bits(si, g["tab_si"]) and same_bits(co, g["tab_co"])


@pytest.mark.parametrize("name", ["mit", "fr079", "intel", "indoor"])
def test_carmen_reader(name):
    g = load(name)
    laser, fn, tag = SETS[name]
    o = Oracle(laser)
    r, x, y, bad, pose = o.read_carmen(os.path.join(GOLDEN, fn), tag)
    assert same_bits(r, g["raw_r"])
    assert np.array_equal(bad, g["raw_bad"])
    if "reader_pose" in g:
        assert same_bits(pose, g["reader_pose"])
        npts = r.shape[1]
        # the padded last beam of (points-1)-beam logs keeps stale x,y in the reference
        assert same_bits(x[:, :npts - 1], g["reader_x"][:, :npts - 1])
        assert same_bits(y[:, :npts - 1], g["reader_y"][:, :npts - 1])
    gp = golden_param(g)
    od = o.param_dict()
    assert od["max_range"] == int(gp["max_range"])
    assert np.float32(od["fi_min"]) == np.float32(gp["fi_min"])
    assert np.float32(od["fi_max"]) == np.float32(gp["fi_max"])


@pytest.mark.parametrize("name", list(SETS))
def test_preprocess(name):
    g = load(name)
    o = make_oracle(name, g)
    for k in range(g["raw_r"].shape[0]):
        r, bad, seg = o.preprocess(g["raw_r"][k], g["raw_bad"][k])
        assert same_bits(r, g["pre_r"][k]) and np.array_equal(bad, g["pre_bad"][k]) and np.array_equal(seg, g["pre_seg"][k])
    for k in range(g["median_only"].shape[0]):
        assert same_bits(o.median(g["raw_r"][k]), g["median_only"][k])


@pytest.mark.parametrize("name", list(SETS))
def test_stages(name):
    g = load(name)
    o = make_oracle(name, g)
    for t, k in enumerate(g["stage_idx"]):
        act = (g["pre_r"][k], g["pre_bad"][k], g["pre_seg"][k])
        nr, nb = o.scan_project(*act, g["stage_pose"][t])
        assert same_bits(nr, g["stage_new_r"][t]) and np.array_equal(nb, g["stage_new_bad"][t])
        st, dth = o.orientation_search(g["pre_r"][k - 1], g["pre_bad"][k - 1], nr, nb)
        assert (st != 0) == (g["stage_orient"][t, 0] != 0)
        if st == 0:
            assert np.float32(dth) == g["stage_orient"][t, 1]
        st, dx, dy, err = o.translation_estimation(g["pre_r"][k - 1], g["pre_bad"][k - 1], nr, nb)
        assert (st != 0) == (g["stage_trans"][t, 0] != 0)
        if st == 0:
            assert same_bits(np.array([dx, dy, err], np.float32), g["stage_trans"][t, 1:])


@pytest.mark.parametrize("name", list(SETS))
@pytest.mark.parametrize("which", [0, 1])
def test_match(name, which):
    g = load(name)
    o = make_oracle(name, g)
    nm = "psm" if which == 0 else "icp"
    S = g["pre_r"].shape[0]
    scans = [(g["pre_r"][k], g["pre_bad"][k], g["pre_seg"][k]) for k in range(S)]
    for b in range(S - 1):
        st, pose, ret, _ = o.match(which, scans[b], scans[b + 1])
        assert (st != 0) == (g[nm + "_status"][b] != 0), b
        assert same_bits(pose, g[nm + "_pose"][b]), (b, pose, g[nm + "_pose"][b])
        if st == 0:
            assert same_bits(np.float32(ret), g[nm + "_ret"][b])
            exy, na = o.error_index2(scans[b], scans[b + 1], (0, 0, 0), pose)
            assert na == g[nm + "_nassoc"][b]
            if na > 0:
                assert same_bits(exy, g[nm + "_exy"][b])
    for b in range(g["init_pose"].shape[0]):
        ip = g["init_pose"][b]
        st, pose, _, _ = o.match(which, scans[b], scans[b + 1], ip[:3], ip[3:])
        assert (st != 0) == (g[nm + "_init_status"][b] != 0)
        assert same_bits(pose, g[nm + "_init_pose_out"][b])


@pytest.mark.parametrize("name", list(SETS))
def test_fmatch_chain(name):
    g = load(name)
    o = make_oracle(name, g)
    n = g["raw_r"].shape[1]
    cur = [g["raw_r"][0].copy(), g["raw_bad"][0].copy(), np.zeros(n, np.int32)]
    for b in range(g["fmatch_pose"].shape[0]):
        act = [g["raw_r"][b + 1].copy(), g["raw_bad"][b + 1].copy(), np.zeros(n, np.int32)]
        pose, pair, _, a2 = o.fmatch(cur, act, b == 0)
        assert same_bits(pose, g["fmatch_pose"][b]), b
        assert same_bits(pair, g["fmatch_pair"][b]), (b, pair, g["fmatch_pair"][b])
        cur = a2


@pytest.mark.parametrize("name", ["mit", "fr079"])
def test_pmap(name):
    z = np.load(os.path.join(GOLDEN, "pmap_%s.npz" % name))
    G = z["geom"]
    m = OraclePMap(int(G[0]), int(G[1]), int(G[2]), int(G[3]), float(G[4]), float(G[5]), float(G[6]))
    for k in range(z["gx"].shape[0]):
        nv = int(z["robot"][k, 2])
        m.update(z["gx"][k, :nv], z["gy"][k, :nv], float(z["robot"][k, 0]), float(z["robot"][k, 1]))
    m.compute()
    hit, sm, prob = m.layers()
    assert np.array_equal(hit, z["hit"].astype(np.float32))
    assert np.array_equal(sm, z["sum"].astype(np.float32))
    assert same_bits(prob, z["prob"])
    cx, cy = m.cloud(0.95)
    assert same_bits(cx, z["cloud_x"]) and same_bits(cy, z["cloud_y"])


def test_grid_line_quirks():
    # degenerate line: one cell
    assert grid_line(5, 5, 5, 5).tolist() == [[5, 5]]
    # always generated from the smaller major coordinate, never reversed (line.cpp:12,111-119)
    a = grid_line(10, 3, 2, 1)
    assert a[0].tolist() == [2, 1] and a[-1].tolist() == [10, 3]
    b = grid_line(2, 1, 10, 3)
    assert b[0].tolist() == [2, 1] and len(a) == len(b) == 9
    c = grid_line(0, 9, 1, 0)
    assert c[0].tolist() == [1, 0] and c[-1].tolist() == [0, 9] and len(c) == 10


@pytest.mark.parametrize("name,laser", [("fr079", "LMS511"), ("intel", "LMS211"), ("mit", "LMS511")])
def test_oracle_on_whole_log_goldens(name, laser):
    """tests/golden/full_<name>.npz (every pair of the shipped logs, reference-made): the CPU oracle on every 40th
    pair and on every pair the reference failed - pins the file the GPU test of the whole logs relies on."""
    from oracle_lib import Oracle
    z = np.load(os.path.join(GOLDEN, "full_%s.npz" % name))
    r = z["raw_r_u16"].astype(np.float32) if "raw_r_u16" in z.files else z["raw_r"]
    bad = z["raw_bad"].astype(np.int32) if "raw_bad" in z.files else np.zeros(r.shape, np.int32)
    P = dict(zip([str(k) for k in z["param_keys"]], z["param"]))
    o = Oracle(laser)
    o.set_fimin_maxrange(np.float32(P["fi_min"]), int(P["max_range"]))
    B = r.shape[0] - 1
    picks = sorted(set(range(0, B, 40)) | set(np.nonzero(z["icp_status"])[0].tolist()) | set(np.nonzero(z["psm_status"])[0].tolist()))
    for b in picks:
        a, c = o.preprocess(r[b], bad[b]), o.preprocess(r[b + 1], bad[b + 1])
        for which, nm in ((0, "psm"), (1, "icp")):
            st, pose, _, _ = o.match(which, a, c)
            assert (st != 0) == (z[nm + "_status"][b] != 0), (name, nm, b)
            if st == 0:
                assert same_bits(pose, z[nm + "_pose"][b]), (name, nm, b)
