"""GPU: point-to-point ICP (CICPWarpper's arithmetic) against oracle/p2p_oracle.c on the
reference's own map-node clouds (src/zhicp/data/{N}.log + matched_nodes.log, inputs only) and on
synthetic clouds with a known transform.  PARITY UNPINNED (MRPT absent)."""
import os

import numpy as np
import pytest

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def cloud(n):
    a = np.loadtxt(os.path.join(GOLDEN, "zhicp", "%d.log" % n), dtype=np.float32)
    return a[:, 0].copy(), a[:, 1].copy()


@pytest.fixture(scope="module")
def ctx():
    from laser_slam_amd import capi
    return capi.Context(capi.param_preset("LMS211"))


def test_reference_node_clouds(ctx):
    from laser_slam_amd import capi
    from oracle_lib import p2p_icp
    rows = np.loadtxt(os.path.join(GOLDEN, "zhicp", "matched_nodes.log"))
    done = 0
    for n1, n2, _, px, py, pth in rows:
        try:
            rx, ry = cloud(int(n1)); ax, ay = cloud(int(n2))
        except OSError:
            continue
        got = capi.p2p_icp_host(ctx, rx, ry, ax, ay, (px, py, pth))
        want = p2p_icp(rx, ry, ax, ay, (px, py, pth))
        assert got[5] == want[5] and got[6] == want[6], (got[:7], want[:7])          # iterations, pairs
        assert np.allclose(got[:5], want[:5], rtol=1e-9, atol=1e-9)
        assert np.allclose(got[7:13], want[7:13], rtol=1e-6, atol=1e-12)
        done += 1
    assert done >= 4


def test_recovers_known_transform(ctx):
    from laser_slam_amd import capi
    rng = np.random.default_rng(3)
    # an L-shaped wall sampled every 2 cm, seen from a pose 20 cm / 4 degrees away
    t = np.arange(0, 4, 0.02)
    ref = np.concatenate([np.stack([t, np.zeros_like(t)], 1), np.stack([np.zeros_like(t), t], 1)]).astype(np.float32)
    th, tx, ty = np.deg2rad(4.0), 0.2, -0.15
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    cur = ((ref - [tx, ty]) @ R).astype(np.float32) + rng.normal(0, 0.002, ref.shape).astype(np.float32)
    out = capi.p2p_icp_host(ctx, ref[:, 0], ref[:, 1], cur[:, 0], cur[:, 1], (0, 0, 0))
    assert abs(out[0] - tx) < 0.02 and abs(out[1] - ty) < 0.02 and abs(out[2] - th) < 0.01
    assert out[3] > 0.8 and out[12] > 0                    # goodness, phi variance
    # empty clouds are handled
    e = capi.p2p_icp_host(ctx, ref[:0, 0], ref[:0, 1], cur[:, 0], cur[:, 1], (0, 0, 0))
    assert e[3] == 0 and e[6] == 0
