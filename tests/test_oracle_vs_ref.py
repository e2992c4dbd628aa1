"""CPU, build container only: the oracle restatement against the REFERENCE compiled in
place (oracle/_ref) on inputs beyond the committed goldens (random poses, non-consecutive
pairs, random rays).  Skipped where oracle/_ref did not travel."""
import numpy as np
import pytest

from helpers import SETS, golden_param, load, same_bits
from oracle_lib import SYN1081, Oracle, OraclePMap, Ref, RefPMap, have_ref

pytestmark = pytest.mark.skipif(not have_ref(), reason="oracle/_ref not built (no /root/reference here)")


def engines(name, g):
    laser = SETS[name][0]
    o = Oracle(laser) if laser else Oracle(**SYN1081)
    r = Ref(laser) if laser else Ref(**SYN1081)
    gp = golden_param(g)
    o.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
    r.set_fimin_maxrange(float(np.float32(gp["fi_min"])), int(gp["max_range"]))
    return o, r


@pytest.mark.parametrize("name", ["mit", "intel", "syn1081"])
def test_random_pairs_and_poses(name):
    g = load(name)
    o, r = engines(name, g)
    rng = np.random.default_rng(11)
    S = g["pre_r"].shape[0]
    sc = lambda k: (g["pre_r"][k], g["pre_bad"][k], g["pre_seg"][k])
    for _ in range(40):
        a = int(rng.integers(0, S)); b = int(np.clip(a + rng.integers(-5, 6), 0, S - 1))
        ip = np.concatenate([rng.normal(0, 40, 2), rng.normal(0, 0.5, 1)]).astype(np.float32)
        for which in (0, 1):
            so, po, _, _ = o.match(which, sc(a), sc(b), (0, 0, 0), ip)
            sr, pr, _ = r.match(which, sc(a), sc(b), (0, 0, 0), ip)
            assert (so != 0) == (sr != 0) and same_bits(po, pr)
        nr_o, nb_o = o.scan_project(*sc(b), ip)
        nr_r, nb_r = r.scan_project(*sc(b), ip)
        assert same_bits(nr_o, nr_r) and np.array_equal(nb_o, nb_r)


def test_pose_algebra():
    o = Oracle("LMS511"); r = Ref("LMS511")
    rng = np.random.default_rng(2)
    for _ in range(50):
        a, b = rng.normal(0, 5, 3), rng.normal(0, 5, 3)
        assert np.array_equal(o.oplus(a, b), r.oplus(a, b))
        assert np.array_equal(o.ominus(a, b), r.ominus(a, b))


def test_pmap_random():
    rng = np.random.default_rng(1)
    m = RefPMap(256, 192, 100, 60, 0.75, -1.5, 0.04)
    o = OraclePMap(256, 192, 100, 60, 0.75, -1.5, 0.04)
    for s in range(30):
        rx, ry = [float(np.float32(v)) for v in rng.uniform(-5, 7, 2)]
        ang = rng.uniform(0, 2 * np.pi, 181); d = rng.uniform(0, 10, 181)
        gx = (rx + d * np.cos(ang)).astype(np.float32); gy = (ry + d * np.sin(ang)).astype(np.float32)
        m.update(gx, gy, rx, ry); o.update(gx, gy, rx, ry)
    m.compute(); o.compute()
    for a, b in zip(m.layers(), o.layers()):
        assert same_bits(a, b)
