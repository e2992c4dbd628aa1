"""GPU: point-to-line ICP (CCanonicalMatcher's arithmetic) against oracle/plicp_oracle.c.
PARITY UNPINNED (CSM absent): the oracle restates the published algorithm with the reference's options;
device and host differ only in the order of the normal-equation sums and in libm vs ocml last bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_beams", [361, 1081])
def test_plicp_matches_oracle_and_recovers_motion(n_beams):
    import torch
    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")
    from laser_slam_amd import capi, synth
    from oracle_lib import plicp
    kw = dict(points=n_beams, fov_deg=270, max_range=3000, min_valid=n_beams // 4, window=n_beams // 6,
              fi_min_deg=-135, fi_max_deg=135)
    p = capi.param_make(**kw)
    ctx = capi.Context(p)
    raw, poses = synth.make_scans(10, n_beams=n_beams)
    raw[3, 40:60] = 3001.0                      # a block of out-of-range readings
    raw[6, ::7] = 5.0                           # readings below the minimum range
    S = raw.shape[0]
    ref_idx = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 2], np.int32); act_idx = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 3, 0], np.int32)
    B = len(ref_idx)
    d_r = torch.from_numpy(raw).cuda()
    d_ri, d_ai = torch.from_numpy(ref_idx).cuda(), torch.from_numpy(act_idx).cuda()
    d_out = torch.zeros(B, 8, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    capi.plicp_batch(ctx, B, d_r, d_ri, d_ai, d_out)
    ctx.sync()
    got = d_out.cpu().numpy()
    for b in range(B):
        want = plicp(raw[ref_idx[b]], raw[act_idx[b]], p.fi_min, p.dfi, p.min_range, p.max_range)
        assert got[b, 3] == want[3] == 1.0
        assert np.allclose(got[b, :3], want[:3], atol=2e-6), (b, got[b], want)
        assert abs(got[b, 6] - want[6]) <= 2 and abs(got[b, 4] - want[4]) <= 1e-3 * want[4] + 1e-6
        # the true motion of the synthetic robot
        x0, y0, t0 = poses[ref_idx[b]]; x1, y1, t1 = poses[act_idx[b]]
        c, s = np.cos(-t0), np.sin(-t0)
        true = np.array([c * (x1 - x0) - s * (y1 - y0), s * (x1 - x0) + c * (y1 - y0), t1 - t0])
        assert np.abs(got[b, :2] - true[:2]).max() < 0.03 and abs(got[b, 2] - true[2]) < 0.01
    # host entry point
    one = capi.plicp_host(ctx, raw[0], raw[1])
    assert np.array_equal(one, got[0])
    ctx.close()


def test_plicp_restart_path():
    """Noisier scans (3 cm) push the mean error over 1 cm: the loop is rerun from six perturbed starts and the
    lowest total error wins - device against oracle on the whole record."""
    import torch
    from laser_slam_amd import capi, synth
    from oracle_lib import plicp
    n_beams = 541
    kw = dict(points=n_beams, fov_deg=270, max_range=3000, min_valid=100, window=50, fi_min_deg=-135, fi_max_deg=135)
    p = capi.param_make(**kw)
    ctx = capi.Context(p)
    raw, _ = synth.make_scans(8, n_beams=n_beams, sigma_m=0.03)
    B = 7
    ri = np.arange(B, dtype=np.int32); ai = ri + 1
    d_out = torch.zeros(B, 8, dtype=torch.float64, device="cuda")
    d_r = torch.from_numpy(raw).cuda()
    torch.cuda.synchronize()
    capi.plicp_batch(ctx, B, d_r, torch.from_numpy(ri).cuda(), torch.from_numpy(ai).cuda(), d_out)
    ctx.sync()
    got = d_out.cpu().numpy()
    restarted = 0
    for b in range(B):
        want = plicp(raw[b], raw[b + 1], p.fi_min, p.dfi, p.min_range, p.max_range)
        assert got[b, 3] == want[3] == 1.0
        assert got[b, 5] == want[5]                                   # same number of iterations over all the loops
        assert np.allclose(got[b, :3], want[:3], atol=5e-6) and abs(got[b, 4] - want[4]) <= 1e-6 * want[4]
        restarted += want[5] > 10
    assert restarted >= B // 2
    # the host entry point runs a restarting pair's six restarts on six workgroups at once and picks the winner itself:
    # the same bits as the one-workgroup form of the batch
    for b in range(B):
        one = capi.plicp_host(ctx, raw[b], raw[b + 1])
        assert np.array_equal(one, got[b]), (b, one, got[b])
    # ... and with the covariance (FMatchKeyFrame2): the restarting pair's is formed at the winner by a phase of its own
    d_out2 = torch.zeros(B, 8, dtype=torch.float64, device="cuda"); d_cov = torch.zeros(B, 9, dtype=torch.float64, device="cuda")
    capi.plicp_batch_cov(ctx, B, d_r, torch.from_numpy(ri).cuda(), torch.from_numpy(ai).cuda(), 0.04, d_out2, d_cov)
    ctx.sync()
    got2, cov = d_out2.cpu().numpy(), d_cov.cpu().numpy()
    assert np.array_equal(got2, got)
    for b in range(B):
        one, c1 = capi.plicp_host_cov(ctx, raw[b], raw[b + 1], 0.04)
        assert np.array_equal(one, got[b]), (b, one, got[b])
        assert np.allclose(c1.reshape(9), cov[b], rtol=1e-9, atol=0), (b, c1, cov[b])
    ctx.close()


@pytest.mark.parametrize("name", ["mit", "fr079", "intel", "indoor"])
def test_plicp_log_pairs_vs_oracle(name):
    """Every pair of the shipped logs' heads up to 6 scans apart.  Range-quantised scans hold many exactly symmetric
    neighbour pairs (equal readings on the rays either side of the match) whose nearer-of-two choice hangs on the
    last bit of the Cartesian points: the device takes the bearings' cos / sin from the host's libm for that reason,
    and the whole record then follows the oracle (same iteration counts, poses within the summation-order noise)."""
    import torch
    torch.zeros(1, device="cuda")
    from laser_slam_amd import capi
    from helpers import SETS, golden_param, load
    from oracle_lib import plicp
    laser, _, _ = SETS[name]
    g = load(name); gp = golden_param(g)
    p = capi.param_preset(laser)
    p.fi_min = float(np.float32(gp["fi_min"])); p.fi_max = float(np.float32(gp["fi_max"])); p.max_range = int(gp["max_range"])
    raw = np.ascontiguousarray(g["raw_r"], np.float32)
    S = raw.shape[0]
    pairs = [(i, j) for i in range(S) for j in range(max(0, i - 6), min(S, i + 7))]
    ri = np.array([a for a, _ in pairs], np.int32); ai = np.array([b for _, b in pairs], np.int32)
    B = len(pairs)
    ctx = capi.Context(p)
    d_out = torch.zeros(B, 8, dtype=torch.float64, device="cuda")
    d_r = torch.from_numpy(raw).cuda()
    torch.cuda.synchronize()
    capi.plicp_batch(ctx, B, d_r, torch.from_numpy(ri).cuda(), torch.from_numpy(ai).cuda(), d_out)
    ctx.sync()
    got = d_out.cpu().numpy()
    restarted = 0
    for k in range(B):
        want = plicp(raw[ri[k]], raw[ai[k]], p.fi_min, p.dfi, p.min_range, p.max_range)
        assert got[k, 3] == want[3], (k, got[k], want)
        assert got[k, 5] == want[5], (k, got[k], want)
        if want[3] == 1.0:
            assert np.abs(got[k, :3] - want[:3]).max() <= 5e-6, (k, got[k], want)
            assert got[k, 6] == want[6] and abs(got[k, 4] - want[4]) <= 1e-6 * want[4] + 1e-9, (k, got[k], want)
            restarted += want[5] > 10
    assert restarted > 0
    ctx.close()


def test_plicp_follows_parameter_changes():
    """The bearing table of the point-to-line ICP is cached per context: after lsa_ctx_set_param (what the CARMEN
    readers' fi_min override ends in) it must be rebuilt - device against the oracle before and after the change."""
    import torch
    torch.zeros(1, device="cuda")
    from laser_slam_amd import capi, synth
    from oracle_lib import plicp
    n_beams = 361
    kw = dict(points=n_beams, fov_deg=270, max_range=3000, min_valid=90, window=60, fi_min_deg=-135, fi_max_deg=135)
    p = capi.param_make(**kw)
    ctx = capi.Context(p)
    raw, _ = synth.make_scans(4, n_beams=n_beams)
    a = capi.plicp_host(ctx, raw[0], raw[1])
    wa = plicp(raw[0], raw[1], p.fi_min, p.dfi, p.min_range, p.max_range)
    assert a[3] == wa[3] == 1.0 and np.allclose(a[:3], wa[:3], atol=5e-6)
    p2 = capi.param_make(**dict(kw, fi_min_deg=-90, fi_max_deg=180))          # the same scans read with another zero bearing
    ctx.set_param(p2)
    b = capi.plicp_host(ctx, raw[0], raw[1])
    wb = plicp(raw[0], raw[1], p2.fi_min, p2.dfi, p2.min_range, p2.max_range)
    assert b[3] == wb[3] == 1.0 and np.allclose(b[:3], wb[:3], atol=5e-6)
    assert np.abs(wb[:2] - wa[:2]).max() > 1e-3                                # the rotation of the frame is visible in the translation
    ctx.close()


@pytest.mark.parametrize("n_beams", [361, 1081])
def test_plicp_covariance_matches_oracle(n_beams):
    """sm_icp's covariance (Censi's closed form) of the device against oracle/plicp_oracle.c's - which
    tests/test_oracle_plicp.py holds to a finite-difference Hessian - incl. pairs that go through the restart path
    (correspondences recomputed at the result) and a failing pair (zeros)."""
    import torch
    from laser_slam_amd import capi, synth
    from oracle_lib import plicp_cov
    kw = dict(points=n_beams, fov_deg=270, max_range=3000, min_valid=n_beams // 4, window=n_beams // 6,
              fi_min_deg=-135, fi_max_deg=135)
    p = capi.param_make(**kw)
    ctx = capi.Context(p)
    raw, _ = synth.make_scans(12, n_beams=n_beams)
    raw = np.concatenate([raw, np.full((1, n_beams), 3001.0, np.float32)], 0)          # a blind scan: the match fails
    ri = np.array([0, 1, 2, 0, 3, 5, 0], np.int32); ai = np.array([1, 2, 3, 4, 9, 11, 12], np.int32)
    B = len(ri)
    d_r = torch.from_numpy(raw).cuda()
    d_out = torch.zeros(B, 8, dtype=torch.float64, device="cuda"); d_cov = torch.full((B, 9), 7.0, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    capi.plicp_batch_cov(ctx, B, d_r, torch.from_numpy(ri).cuda(), torch.from_numpy(ai).cuda(), 0.04, d_out, d_cov)
    ctx.sync()
    got, cov = d_out.cpu().numpy(), d_cov.cpu().numpy().reshape(B, 3, 3)
    restarted = 0
    for b in range(B):
        want, wcov, _ = plicp_cov(raw[ri[b]], raw[ai[b]], p.fi_min, p.dfi, p.min_range, p.max_range, 0.04)
        assert got[b, 3] == want[3]
        if want[3] == 0:
            assert np.all(cov[b] == 0)
            continue
        assert np.allclose(got[b, :3], want[:3], atol=2e-6)
        restarted += want[5] > 10
        # same correspondences, same formulas; sums in another order and ocml against libm in the trigonometry
        assert np.allclose(cov[b], wcov, rtol=1e-6, atol=1e-9 * np.abs(wcov).max()), (b, cov[b], wcov)
        assert np.all(np.linalg.eigvalsh(0.5 * (cov[b] + cov[b].T)) > 0)
    assert got[B - 1, 3] == 0
    one, c1 = capi.plicp_host_cov(ctx, raw[0], raw[1], 0.04)
    assert np.array_equal(one, got[0]) and np.allclose(c1, cov[0], rtol=1e-9, atol=0)
    ctx.close()
