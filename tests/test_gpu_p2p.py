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
    import torch
    assert torch.cuda.is_available()       # torch first: it refuses to initialise HIP after another library has
    torch.zeros(1, device="cuda")
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


def test_map_node_rasterise_cloud_icp_chain(ctx):
    """The one place where the reference chains CPMap and CICPWarpper (CMapNode::computePMAP +
    matchNodePairICP, src/mapGraph/MapNode.cpp:703-757, 625-656): the scans of a map node are
    rasterised into a node-local counting grid, the cells above the threshold become the node's
    point cloud, and two nodes' clouds are aligned by point-to-point ICP.  Device chain (batched grid
    update -> probabilities -> cloud -> ICP) against the oracle chain; the clouds must be identical."""
    import torch
    from laser_slam_amd import capi, synth
    from oracle_lib import OraclePMap, p2p_icp
    n_beams, fov = 361, 270.0
    raw, poses = synth.make_scans(30, n_beams=n_beams, fov_deg=fov)
    bearings = np.deg2rad(-fov / 2 + fov * np.arange(n_beams) / (n_beams - 1))

    def node(first, last):
        """scans first..last-1 in the frame of scan `first` (metres), as computeGlobalScans does"""
        x0, y0, t0 = poses[first]
        c, s = np.cos(-t0), np.sin(-t0)
        gx, gy, rob = [], [], []
        for k in range(first, last):
            px, py, pt = poses[k]
            lx, ly = c * (px - x0) - s * (py - y0), s * (px - x0) + c * (py - y0)
            lt = pt - t0
            r = raw[k] / 100.0
            gx.append((lx + r * np.cos(lt + bearings)).astype(np.float32))
            gy.append((ly + r * np.sin(lt + bearings)).astype(np.float32))
            rob.append((np.float32(lx), np.float32(ly)))
        gx, gy, rob = np.stack(gx), np.stack(gy), np.array(rob, np.float32)
        # computeBoundary / computePMAP: grid twice the extent, origin in the middle, 5 cm cells
        size_x = int((gx.max() - gx.min() + 0.05) / 0.05) * 2
        size_y = int((gy.max() - gy.min() + 0.05) / 0.05) * 2
        grid = capi.Grid(ctx, size_x, size_y, size_x // 2, size_y // 2, 0.0, 0.0, 0.05)
        torch.cuda.synchronize()
        grid.update_batch(len(gx), n_beams, torch.from_numpy(gx).cuda(), torch.from_numpy(gy).cuda(), torch.from_numpy(rob).cuda())
        grid.compute_prob()
        cx_, cy_ = grid.cloud(0.5)
        o = OraclePMap(size_x, size_y, size_x // 2, size_y // 2, 0.0, 0.0, 0.05)
        for k in range(len(gx)):
            o.update(gx[k], gy[k], float(rob[k, 0]), float(rob[k, 1]))
        o.compute()
        ox, oy = o.cloud(0.5)
        assert np.array_equal(cx_, ox) and np.array_equal(cy_, oy) and len(ox) > 200
        grid.close()
        return cx_, cy_

    ax, ay = node(0, 12)
    bx, by = node(6, 18)
    # true pose of node B's frame in node A's frame
    x0, y0, t0 = poses[0]; x1, y1, t1 = poses[6]
    c, s = np.cos(-t0), np.sin(-t0)
    true = np.array([c * (x1 - x0) - s * (y1 - y0), s * (x1 - x0) + c * (y1 - y0), t1 - t0])
    init = true + [0.05, -0.04, 0.02]                  # the odometry guess the graph hands to matchNodePairICP
    got = capi.p2p_icp_host(ctx, ax, ay, bx, by, init)
    want = p2p_icp(ax, ay, bx, by, init)
    assert got[5] == want[5] and got[6] == want[6]
    assert np.allclose(got[:5], want[:5], rtol=1e-9, atol=1e-9)
    assert np.abs(got[:2] - true[:2]).max() < 0.05 and abs(got[2] - true[2]) < 0.02 and got[3] > 0.5


@pytest.mark.parametrize("name", ["fr079", "intel"])
def test_consecutive_log_scans(ctx, name):
    """BASELINE configs 1 / 2: CICPWarpper's ICP on consecutive scans of the shipped logs' heads (the reader's
    Cartesian points, cm -> m, readings at the maximum range dropped), zero prior - device against the oracle."""
    from laser_slam_amd import capi
    from helpers import load
    from oracle_lib import p2p_icp
    g = load(name)
    x, y, bad, r = g["reader_x"] * 0.01, g["reader_y"] * 0.01, g["raw_bad"], g["raw_r"]
    mx = float(g["param"][list(g["param_keys"]).index("max_range")]) if "max_range" in list(g["param_keys"]) else 5000.0
    done = 0
    for k in range(0, 40):
        a = (bad[k] == 0) & (r[k] < mx); b = (bad[k + 1] == 0) & (r[k + 1] < mx)
        rx, ry = x[k][a].astype(np.float32), y[k][a].astype(np.float32)
        cx, cy = x[k + 1][b].astype(np.float32), y[k + 1][b].astype(np.float32)
        got = capi.p2p_icp_host(ctx, rx, ry, cx, cy, (0, 0, 0))
        want = p2p_icp(rx, ry, cx, cy, (0, 0, 0))
        assert got[5] == want[5] and got[6] == want[6], (k, got[:7], want[:7])
        assert np.allclose(got[:5], want[:5], rtol=1e-9, atol=1e-9)
        assert np.allclose(got[7:13], want[7:13], rtol=1e-6, atol=1e-12)
        done += got[3] > 0.5
    assert done >= 30                                   # consecutive scans pair up


def test_batch_equals_single_problems(ctx):
    """lsa_p2p_icp_batch (what bench.py's zhicp entries time) over concatenated clouds with offsets = the same problems one
    by one through lsa_p2p_icp_host, bit for bit - ragged sizes, an empty current cloud and non-zero priors included."""
    import torch
    from laser_slam_amd import capi
    rng = np.random.default_rng(8)
    probs = []
    for k in range(9):
        n_ref, n_cur = int(rng.integers(30, 700)), (0 if k == 4 else int(rng.integers(30, 700)))
        t = np.sort(rng.uniform(0, 6, n_ref))
        ref = np.stack([t, 0.3 * np.sin(2 * t) + (t > 3) * 0.8], 1).astype(np.float32)
        th, tx, ty = rng.uniform(-0.08, 0.08), rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2)
        u = np.sort(rng.uniform(0, 6, n_cur))
        cur_w = np.stack([u, 0.3 * np.sin(2 * u) + (u > 3) * 0.8], 1)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        cur = ((cur_w - [tx, ty]) @ R).astype(np.float32)
        init = (0.0, 0.0, 0.0) if k % 2 else (tx * 0.5, ty * 0.5, th * 0.5)
        probs.append((ref, cur, init))
    rx = np.concatenate([p[0][:, 0] for p in probs]); ry = np.concatenate([p[0][:, 1] for p in probs])
    cx = np.concatenate([p[1][:, 0] for p in probs]); cy = np.concatenate([p[1][:, 1] for p in probs])
    ro = np.concatenate([[0], np.cumsum([len(p[0]) for p in probs])]).astype(np.int32)
    co = np.concatenate([[0], np.cumsum([len(p[1]) for p in probs])]).astype(np.int32)
    init = np.array([p[2] for p in probs], np.float64)
    d = [torch.from_numpy(np.ascontiguousarray(a)).cuda() for a in (rx, ry, ro, cx, cy, co, init)]
    out = torch.zeros(len(probs), 16, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    capi.p2p_icp_batch(ctx, len(probs), d[0], d[1], d[2], d[3], d[4], d[5], len(rx), max(len(cx), 1), d[6], out)
    got = out.cpu().numpy()
    for k, (ref, cur, ini) in enumerate(probs):
        want = capi.p2p_icp_host(ctx, ref[:, 0].copy(), ref[:, 1].copy(), cur[:, 0].copy(), cur[:, 1].copy(), ini)
        assert np.array_equal(got[k, :13], want[:13]), (k, got[k, :7], want[:7])
    assert got[4, 3] == 0 and got[4, 6] == 0


def test_pairs_do_not_depend_on_the_sort_axis(ctx, monkeypatch):
    """The nearest-neighbour search scans the first map sorted along an axis the kernel picks per problem among eight
    directions.  The search is exact on any of them: every forced axis (LSA_P2P_AXIS, a test hook read at launch) must give
    the bits the automatic choice gives - node clouds of map cells (columns of equal x), scans, ragged synthetic clouds."""
    from laser_slam_amd import capi
    rng = np.random.default_rng(21)
    probs = []
    rows = np.loadtxt(os.path.join(GOLDEN, "zhicp", "matched_nodes.log"))
    for n1, n2, _, px, py, pth in rows[:3]:
        try:
            probs.append((cloud(int(n1)), cloud(int(n2)), (px, py, pth)))
        except OSError:
            pass
    for k in range(4):
        n_ref, n_cur = int(rng.integers(130, 900)), int(rng.integers(130, 900))
        t = np.sort(rng.uniform(0, 8, n_ref)); u = np.sort(rng.uniform(0, 8, n_cur))
        wall = lambda v: np.stack([np.where(v < 4, v, 4.0), np.where(v < 4, 0.02 * np.sin(9 * v), v - 4.0)], 1)      # a corner: one wall along x, one along y
        th, tx, ty = rng.uniform(-0.06, 0.06), rng.uniform(-0.15, 0.15), rng.uniform(-0.15, 0.15)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        ref = wall(t).astype(np.float32); cur = ((wall(u) - [tx, ty]) @ R).astype(np.float32)
        probs.append(((ref[:, 0].copy(), ref[:, 1].copy()), (cur[:, 0].copy(), cur[:, 1].copy()), (0.0, 0.0, 0.0)))
    assert len(probs) >= 5
    monkeypatch.delenv("LSA_P2P_AXIS", raising=False)
    auto = [capi.p2p_icp_host(ctx, r[0], r[1], c[0], c[1], ini) for r, c, ini in probs]
    assert any(a[6] > 50 for a in auto)
    for axis in range(8):
        monkeypatch.setenv("LSA_P2P_AXIS", str(axis))
        for (r, c, ini), want in zip(probs, auto):
            got = capi.p2p_icp_host(ctx, r[0], r[1], c[0], c[1], ini)
            assert np.array_equal(got[:13], want[:13]), (axis, got[:7], want[:7])


def test_degenerate_clouds_against_oracle(ctx):
    """Clouds the axis choice and the strips of the sort could trip over: all points on one vertical or horizontal line (the
    projections' extent vanishes on some axes), every point the same, exactly 128 points (the threshold of the choice), a
    first map of 2048 points (the largest that is sorted) and one of 2049 (streamed in tiles, unsorted)."""
    from laser_slam_amd import capi
    from oracle_lib import p2p_icp
    rng = np.random.default_rng(33)
    def moved(p, th, tx, ty):
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        return ((p - [tx, ty]) @ R).astype(np.float32)
    t = np.linspace(0, 5, 300)
    cases = {
        "vertical": np.stack([np.full(300, 1.5), t], 1),
        "horizontal": np.stack([t, np.full(300, -0.7)], 1),
        "same point": np.tile([[0.4, 0.9]], (200, 1)),
        "128 points": np.stack([t[:128], 0.2 * np.sin(3 * t[:128])], 1),
        "2048 points": np.stack([np.linspace(0, 20, 2048), np.sin(np.linspace(0, 20, 2048))], 1),
        "2049 points": np.stack([np.linspace(0, 20, 2049), np.sin(np.linspace(0, 20, 2049))], 1),
    }
    for name, ref in cases.items():
        ref = ref.astype(np.float32)
        cur = moved(ref[:: 2 if len(ref) > 1000 else 1].astype(np.float64) + rng.normal(0, 0.002, (len(ref[:: 2 if len(ref) > 1000 else 1]), 2)), 0.02, 0.05, -0.03)
        got = capi.p2p_icp_host(ctx, ref[:, 0].copy(), ref[:, 1].copy(), cur[:, 0].copy(), cur[:, 1].copy(), (0.0, 0.0, 0.0))
        want = p2p_icp(ref[:, 0].copy(), ref[:, 1].copy(), cur[:, 0].copy(), cur[:, 1].copy(), (0.0, 0.0, 0.0))
        assert got[5] == want[5] and got[6] == want[6], (name, got[:7], want[:7])
        assert np.allclose(got[:5], want[:5], rtol=1e-9, atol=1e-9), (name, got[:5], want[:5])
