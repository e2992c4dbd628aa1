"""GPU: likelihood field, pose scoring, arg-max, weights and systematic resampling against
oracle/pose_oracle.c.  PARITY UNPINNED (the reference's localization module needs Qt4 / Boost /
MRPT and cannot be built): the oracle restates the readable source; tolerances cover exp/log
implementation differences only."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def world():
    import torch
    from laser_slam_amd import capi, synth
    assert torch.cuda.is_available()
    rng = np.random.default_rng(0)
    sx, sy, res = 260, 200, 0.05
    prob = np.full((sx, sy), 0.1, np.float32)              # free space
    prob[:, :3] = 0.9; prob[:, -3:] = 0.9; prob[:3, :] = 0.9; prob[-3:, :] = 0.9     # walls
    prob[100:104, 60:140] = 0.95                                                     # a wall segment
    prob[180:200, 20:40] = 0.5                                                       # unknown block
    prob[40:60, 150:170] = 0.45                                                      # "unknown-ish" (|p-0.5| <= 0.1)
    prob += rng.uniform(0, 0.02, prob.shape).astype(np.float32)
    ctx = capi.Context(capi.param_preset("LMS151"))
    fld = capi.Field(ctx, sx, sy, 0, 0, 0.0, 0.0, res)
    d_prob = torch.from_numpy(prob).cuda()
    torch.cuda.synchronize()
    fld.build(d_prob, 1.0, 0.3, 0.5)
    ctx.sync()
    return dict(torch=torch, capi=capi, ctx=ctx, fld=fld, prob=prob, res=res, rng=rng)


def test_field_matches_oracle(world):
    from oracle_lib import likelihood_field
    lik_o = likelihood_field(world["prob"], world["res"])
    lik_g = world["fld"].download(0)
    assert np.allclose(lik_g, lik_o, rtol=2e-6, atol=1e-9)
    # exact where it matters: the windowed distance transform picks the same nearest cell
    assert np.array_equal(lik_g == lik_g.max(), lik_o == lik_o.max())
    log_g = world["fld"].download(1)
    assert np.allclose(log_g, np.log(lik_o.astype(np.float64)), rtol=0, atol=2e-6)


def test_scores_argmax_weights_resample(world):
    torch, capi, ctx, fld = world["torch"], world["capi"], world["ctx"], world["fld"]
    from oracle_lib import likelihood_field, neff, score_poses, systematic_resample
    rng = world["rng"]
    n = 541
    ang = np.linspace(-0.75 * np.pi, 0.75 * np.pi, n)
    rr = rng.uniform(0.5, 6.0, n)
    scan_x = (rr * np.cos(ang)).astype(np.float32); scan_y = (rr * np.sin(ang)).astype(np.float32)
    P = 5000
    poses = np.stack([rng.uniform(-1, 14, P), rng.uniform(-1, 11, P), rng.uniform(-7, 7, P)], 1).astype(np.float32)
    d_x, d_y, d_p = (torch.from_numpy(a).cuda() for a in (scan_x, scan_y, poses))
    d_s = torch.zeros(P, device="cuda")
    torch.cuda.synchronize()          # the context launches on its own stream
    fld.score_poses(d_x, d_y, n, 3, d_p, P, d_s)
    ctx.sync()
    got = d_s.cpu().numpy()
    lik_o = likelihood_field(world["prob"], world["res"])
    want = score_poses(lik_o, 0, 0, 0.0, 0.0, world["res"], scan_x, scan_y, 3, poses)
    assert np.allclose(got, want, rtol=1e-5, atol=2e-3), np.abs(got - want).max()
    # the plain pose (+) point frame of MRPT-style callers
    fld.score_poses(d_x, d_y, n, 3, d_p, P, d_s, std_frame=True)
    ctx.sync()
    want_std = score_poses(lik_o, 0, 0, 0.0, 0.0, world["res"], scan_x, scan_y, 3, poses, std_frame=True)
    assert np.allclose(d_s.cpu().numpy(), want_std, rtol=1e-5, atol=2e-3)
    assert not np.allclose(want_std, want, atol=1.0)
    # decimation 1 (np < 100 rule of the reference is the caller's choice here)
    fld.score_poses(d_x, d_y, 90, 1, d_p, 64, d_s)
    ctx.sync()
    want1 = score_poses(lik_o, 0, 0, 0.0, 0.0, world["res"], scan_x[:90], scan_y[:90], 1, poses[:64])
    assert np.allclose(d_s.cpu().numpy()[:64], want1, rtol=1e-5, atol=1e-3)
    fld.score_poses(d_x, d_y, n, 3, d_p, P, d_s)
    ctx.sync()
    m, i = capi.pose_argmax(ctx, d_s, P)
    assert m == got.max() and i == int(np.argmax(got))
    # weights / neff
    d_w = torch.zeros(P, device="cuda")
    torch.cuda.synchronize()
    s, s2 = capi.pf_weights(ctx, d_s, P, m, d_w)
    w = d_w.cpu().numpy()
    assert np.allclose(w, np.exp(got - m), rtol=2e-6, atol=1e-30)
    assert abs(s - w.astype(np.float64).sum()) < 1e-9 * s and abs(s2 - (w.astype(np.float64) ** 2).sum()) < 1e-9 * s2
    assert abs(1.0 / (1.0 + s2 / (s * s)) - neff(w)) < 1e-9
    # systematic resampling: all offspring, and a shard of them
    d_par = torch.zeros(P, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    capi.pf_resample(ctx, d_w, P, 0.6180339, P, 0, P, d_par)
    par_o = systematic_resample(w, 0.6180339, P)
    par_g = d_par.cpu().numpy()
    # plain vs Kahan cumulative sums can move a boundary by one particle at most, and only
    # where an offspring lands within rounding of a cumulative weight
    assert (par_g != par_o).mean() < 1e-3 and np.abs(par_g - par_o).max() <= 1
    capi.pf_resample(ctx, d_w, P, 0.6180339, P, 1234, 777, d_par)
    assert np.array_equal(d_par.cpu().numpy()[:777], par_g[1234:1234 + 777])
    d_new = torch.zeros(P, 3, device="cuda")
    torch.cuda.synchronize()
    capi.pf_resample(ctx, d_w, P, 0.6180339, P, 0, P, d_par)
    capi.gather_poses(ctx, d_p, d_par, P, d_new)
    ctx.sync()
    assert np.array_equal(d_new.cpu().numpy(), poses[par_g])


@pytest.mark.parametrize("geom", [(37, 11, -3.2, 1.7, 0.07, 150, 230), (0, 0, 0.0, 0.0, 0.05, 64, 64), (500, 3, 12.5, -40.0, 0.1, 700, 90)])
def test_scorer_map_geometries_and_poses_off_the_map(geom):
    """The scorer's fixed-point cell index, its tiled table with the four-cell apron and the clamp of far-away beams
    against the oracle's plain arithmetic: map origins, offsets and resolutions that are not the fixture's, poses inside,
    on the edges (cells -1, 0, size - 2, size - 1: the reference's `(int)` truncation and its `>= size - 1` test), far
    outside (every beam clamped) and with huge coordinates; both frames; beam counts that leave a ragged last trip."""
    import torch
    from laser_slam_amd import capi
    from oracle_lib import likelihood_field, score_poses
    cx, cy, offx, offy, res, sx, sy = geom
    rng = np.random.default_rng(sx * 7 + sy)
    prob = np.full((sx, sy), 0.1, np.float32)
    prob[:, :2] = 0.9; prob[:, -2:] = 0.9; prob[:2, :] = 0.9; prob[-2:, :] = 0.9
    prob[sx // 3:sx // 3 + 3, sy // 4:3 * sy // 4] = 0.95
    prob += rng.uniform(0, 0.02, prob.shape).astype(np.float32)
    ctx = capi.Context(capi.param_preset("LMS151"))
    fld = capi.Field(ctx, sx, sy, cx, cy, offx, offy, res)
    d_prob = torch.from_numpy(prob).cuda(); torch.cuda.synchronize()
    fld.build(d_prob, 1.0, 0.3, 0.5); ctx.sync()
    lik_o = likelihood_field(prob, res)
    # world coordinates of the map: x in [(0 - cx) res + offx, (sx - cx) res + offx)
    x0, x1 = (0 - cx) * res + offx, (sx - cx) * res + offx
    y0, y1 = (0 - cy) * res + offy, (sy - cy) * res + offy
    P = 3000
    poses = np.stack([rng.uniform(x0, x1, P), rng.uniform(y0, y1, P), rng.uniform(-7, 7, P)], 1)
    poses[:200, 0] = rng.choice([x0 - res, x0, x0 + res * 0.999, x1 - 2 * res, x1 - res, x1], 200) + rng.uniform(-1e-4, 1e-4, 200)
    poses[200:400, 1] = rng.choice([y0 - res, y0, y1 - 2 * res, y1 - res, y1], 200) + rng.uniform(-1e-4, 1e-4, 200)
    poses[400:500, :2] += rng.choice([-1, 1], (100, 2)) * rng.uniform(50, 5000, (100, 2))          # far outside
    poses[500:520, :2] = rng.choice([-1, 1], (20, 2)) * 1e7                                           # huge
    poses = poses.astype(np.float32)
    d_p = torch.from_numpy(poses).cuda(); d_s = torch.zeros(P, device="cuda")
    for n, decim in ((541, 3), (97, 1), (1081, 3), (8, 1)):
        ang = np.linspace(-0.75 * np.pi, 0.75 * np.pi, n)
        rr = rng.uniform(0.05, 0.3 * max(x1 - x0, y1 - y0), n)
        scan_x = (rr * np.cos(ang)).astype(np.float32); scan_y = (rr * np.sin(ang)).astype(np.float32)
        d_x, d_y = torch.from_numpy(scan_x).cuda(), torch.from_numpy(scan_y).cuda()
        for std in (False, True):
            torch.cuda.synchronize()
            fld.score_poses(d_x, d_y, n, decim, d_p, P, d_s, std_frame=std); ctx.sync()
            want = score_poses(lik_o, cx, cy, offx, offy, res, scan_x, scan_y, decim, poses, std_frame=std)
            got = d_s.cpu().numpy()
            assert np.allclose(got, want, rtol=1e-5, atol=2e-3), (n, std, np.abs(got - want).max(), int(np.argmax(np.abs(got - want))))
    fld.close(); ctx.close()


def test_standard_resampler(world):
    """Standard_resampler (SIRFlt.cpp:39-115, what SIR_scheme::update_resample() uses): one uniform per offspring,
    parents ascending as copy_resamples leaves them; device against oracle (the device sums the weights pairwise,
    the oracle with Kahan's recurrence: an offspring on a slice boundary may land on the neighbour)."""
    import torch
    from laser_slam_amd import capi
    from oracle_lib import standard_resample
    ctx = world["ctx"]
    rng = np.random.default_rng(9)
    for P, n_u in ((60, 60), (5000, 5000), (200000, 200000), (1000, 4000)):
        w = (rng.uniform(0, 1, P) ** 6).astype(np.float32)
        w[P // 3: P // 3 + 7] = 0.0
        u = rng.uniform(0, 1, n_u)
        d_w = torch.from_numpy(w).cuda(); d_u = torch.from_numpy(u).cuda()
        d_par = torch.full((n_u,), -1, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        capi.pf_resample_standard(ctx, d_w, P, d_u, n_u, d_par)
        got = d_par.cpu().numpy()
        want, left = standard_resample(w, u)
        assert left == 0
        assert np.all(np.diff(got) >= 0) and got.min() >= 0 and got.max() < P
        assert (got != want).mean() < 1e-3 and np.abs(got - want).max() <= 1
        assert not np.isin(got, np.arange(P // 3, P // 3 + 7)).any()
        # a shard's block of the same list
        first, M = n_u // 3, min(777, n_u - n_u // 3)
        d_blk = torch.full((M,), -1, dtype=torch.int32, device="cuda")
        capi.pf_resample_standard(ctx, d_w, P, d_u, n_u, d_blk, first, M)
        assert np.array_equal(d_blk.cpu().numpy(), got[first:first + M])
    # all the weight on one particle; zero weights are an error like the reference's "zero cumulative weight sum"
    w = np.zeros(100, np.float32); w[42] = 3.0
    d_par = torch.zeros(50, dtype=torch.int32, device="cuda")
    capi.pf_resample_standard(ctx, torch.from_numpy(w).cuda(), 100, torch.from_numpy(rng.uniform(0, 1, 50)).cuda(), 50, d_par)
    assert (d_par.cpu().numpy() == 42).all()
    with pytest.raises(Exception):
        capi.pf_resample_standard(ctx, torch.zeros(100, device="cuda"), 100, torch.from_numpy(rng.uniform(0, 1, 50)).cuda(), 50, d_par)


def test_sharded_filter_single_gpu(world):
    """ShardedParticleFilter wired to the HIP entry points (world size 1)."""
    torch, capi, ctx, fld = world["torch"], world["capi"], world["ctx"], world["fld"]
    from laser_slam_amd import pf
    rng = np.random.default_rng(4)
    n = 361
    ang = np.linspace(-np.pi / 2, np.pi / 2, n)
    scan_x = (3.0 * np.cos(ang)).astype(np.float32); scan_y = (3.0 * np.sin(ang)).astype(np.float32)
    d_x, d_y = torch.from_numpy(scan_x).cuda(), torch.from_numpy(scan_y).cuda()
    P = 20000
    poses = torch.from_numpy(np.stack([rng.uniform(1, 12, P), rng.uniform(1, 9, P), rng.uniform(-3, 3, P)], 1).astype(np.float32)).cuda()

    prims = pf.DevicePrims(ctx, fld, d_x, d_y, n, 3)
    f = pf.ShardedParticleFilter(P, prims, device="cuda")
    new, info = f.step(poses, 0.25)
    ctx.sync()
    assert info["resampled"] and new.shape == (P, 3) and f.bytes_sent == 0
    # resampling concentrates the particles on high-score poses
    assert float(prims.score(new).mean()) > float(prims.score(poses).mean())
    # ... and is systematic resampling of the fixed-point weights by its definition
    s = prims.score(poses); ctx.sync()
    w = np.exp(s.cpu().numpy() - np.float32(info["best_score"])).astype(np.float32)
    wd, _, _ = prims.weights(s, info["best_score"]); ctx.sync()
    q = [int(v) for v in np.floor(np.clip(wd.cpu().numpy().astype(np.float64), 0, 1) * 4294967296.0)]
    W = sum(q); U0 = (int(np.ldexp(0.25, 53)) * W) >> 53
    c, e = 0, []
    for v in q:
        c += v
        x = c * P - U0
        e.append(0 if x <= 0 else min(P, -(-x // W)))
    par = np.searchsorted(np.array(e), np.arange(P), side="right")
    assert np.array_equal(new.cpu().numpy(), poses.cpu().numpy()[par])
    assert np.allclose(wd.cpu().numpy(), w, rtol=2e-6)


def test_scan_simulator_and_beam_model(world):
    """CVPmap::laserScanSimulator / obsLikelyhood2 against oracle/sim_oracle.c.  Ranges are whole
    numbers of ray-march steps; device and host cos/sin/exp may differ in the last bit, so a ray that
    passes within an ulp of a cell boundary may stop one step apart: at most 0.1 % of the rays may
    differ, by one step, and the beam-model scores agree to 1e-3 relative."""
    torch, capi, ctx, fld = world["torch"], world["capi"], world["ctx"], world["fld"]
    from oracle_lib import beam_model, simulate_scan
    rng = np.random.default_rng(11)
    prob, res = world["prob"], world["res"]
    n, P = 181, 96
    poses = np.stack([rng.uniform(0.5, 12.5, P), rng.uniform(0.5, 9.5, P), rng.uniform(-7, 7, P)], 1).astype(np.float32)
    poses[0] = [-3.0, 2.0, 0.1]                     # outside the map: every ray invalid at range 0
    d_prob = torch.from_numpy(prob).cuda(); d_p = torch.from_numpy(poses).cuda()
    d_r = torch.zeros(P, n, device="cuda"); d_b = torch.zeros(P, n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    capi.simulate_scans(fld, d_prob, d_p, P, n, d_r, d_b, max_range=8.0)
    ctx.sync()
    r_g, b_g = d_r.cpu().numpy(), d_b.cpu().numpy()
    r_o = np.zeros((P, n), np.float32); b_o = np.zeros((P, n), np.int32)
    for p in range(P):
        r_o[p], b_o[p] = simulate_scan(prob, 0, 0, 0.0, 0.0, res, poses[p], n, capi.SIM_FOV, capi.SIM_MIN_ANGLE, 8.0)
    diff = (r_g != r_o) | (b_g != b_o)
    assert diff.mean() <= 1e-3
    assert np.abs(r_g - r_o).max() <= 1.01 * res
    assert b_g[0].all() and (r_g[0] == 0).all()
    assert 0.02 < b_g.mean() < 0.9                  # the test world produces both valid and invalid rays
    # beam model: the scan seen from pose 5, scored at all poses
    obs_r, obs_bad = r_o[5].copy(), b_o[5].astype(np.uint8)
    d_or, d_ob = torch.from_numpy(obs_r).cuda(), torch.from_numpy(obs_bad).cuda()
    d_s = torch.zeros(P, device="cuda")
    torch.cuda.synchronize()
    capi.beam_model_scores(fld, d_prob, d_p, P, d_or, d_ob, n, d_s, max_range=8.0)
    ctx.sync()
    s_g = d_s.cpu().numpy()
    s_o = beam_model(prob, 0, 0, 0.0, 0.0, res, poses, obs_r, obs_bad.astype(np.int32), capi.SIM_FOV, capi.SIM_MIN_ANGLE, 8.0)
    assert np.allclose(s_g, s_o, rtol=1e-3, atol=1e-3)
    assert s_g.argmax() == 5 and s_g[5] == np.float32(1 + (obs_bad == 0).sum())


def test_topk_estimate_predict_roughen(world):
    """topK / weightMean / meanDis / predict / roughen_minmax against oracle/sim_oracle.c."""
    torch, capi, ctx = world["torch"], world["capi"], world["ctx"]
    import oracle_lib as ol
    rng = np.random.default_rng(21)
    n = 20000
    score = rng.random(n).astype(np.float32)
    score[[17, 4000, 19999]] = 2.0                  # equal maxima come out in index order
    score[123] = -1.0
    poses = rng.normal(size=(n, 3)).astype(np.float32)
    d_s, d_p = torch.from_numpy(score).cuda(), torch.from_numpy(poses).cuda()
    torch.cuda.synchronize()
    for k in (1, 8, 60):
        idx, sc = capi.pose_topk(ctx, d_s, n, k)
        idx_o = ol.topk(score, k)
        assert np.array_equal(idx, idx_o) and np.array_equal(sc, score[idx_o])
    mean, mw, md, idx = capi.pf_estimate(ctx, d_p, d_s, n, 8)
    mean_o, mw_o, md_o = ol.pf_estimate(poses, score, ol.topk(score, 8))
    assert np.array_equal(mean, mean_o) and mw == mw_o and md == md_o        # same double arithmetic on the host
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    d_n = torch.from_numpy(nrm).cuda(); d_out = torch.zeros(n, 3, device="cuda")
    torch.cuda.synchronize()
    capi.pf_predict(ctx, n, [1.5, -2.0, 0.7], [0.25, 0.25, (np.pi / 6) ** 2], d_n, d_out)
    ctx.sync()
    assert np.array_equal(d_out.cpu().numpy(), ol.pf_predict(n, [1.5, -2.0, 0.7], [0.25, 0.25, (np.pi / 6) ** 2], nrm))
    d_m = torch.from_numpy(poses).cuda()
    torch.cuda.synchronize()
    capi.pf_motion(ctx, n, [0.31, -0.04, 0.11], [0.04, 0.04, 0.035], d_n, d_m)
    ctx.sync()
    assert np.array_equal(d_m.cpu().numpy(), ol.pf_motion(poses, [0.31, -0.04, 0.11], [0.04, 0.04, 0.035], nrm))
    d_q = torch.from_numpy(poses).cuda()
    torch.cuda.synchronize()
    capi.pf_roughen(ctx, n, 0.2, d_n, d_q)
    ctx.sync()
    assert np.array_equal(d_q.cpu().numpy(), ol.pf_roughen(poses, 0.2, nrm))


def test_icp_goodness_scorer(world):
    """CVPmap::obsLikelyhood (the scorer the reference's PF calls): virtual scan + ICPMatch per pose,
    against the oracle composition (sim_oracle.c + p2p_oracle.c).  Tolerances as for the p2p tests."""
    torch, capi, ctx, fld = world["torch"], world["capi"], world["ctx"], world["fld"]
    from oracle_lib import icp_goodness, simulate_scan
    rng = np.random.default_rng(31)
    prob, res = world["prob"], world["res"]
    n = 181
    true = np.array([6.0, 5.0, 0.4], np.float32)
    r, bad = simulate_scan(prob, 0, 0, 0.0, 0.0, res, true, n, capi.SIM_FOV, capi.SIM_MIN_ANGLE, 30.0)
    AA = np.float32(capi.SIM_FOV) / np.float32(n - 1)
    ang = (np.float32(-3.141592654 / 4.0) + np.arange(n) * float(AA)).astype(np.float32)
    obs_x, obs_y = (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)
    P = 24
    poses = (true + np.concatenate([np.zeros((1, 3)), rng.normal(0, [0.15, 0.15, 0.05], (P - 2, 3)), [[-3, 3, 2.0]]])).astype(np.float32)
    d_prob = torch.from_numpy(prob).cuda(); d_p = torch.from_numpy(poses).cuda()
    d_ox, d_oy = torch.from_numpy(obs_x).cuda(), torch.from_numpy(obs_y).cuda()
    d_w = torch.zeros(P, device="cuda"); d_rel = torch.zeros(P, 3, device="cuda")
    torch.cuda.synchronize()
    capi.icp_goodness_scores(fld, d_prob, d_p, P, d_ox, d_oy, n, d_w, d_rel)
    ctx.sync()
    w_g, rel_g = d_w.cpu().numpy(), d_rel.cpu().numpy()
    w_o, rel_o = icp_goodness(prob, 0, 0, 0.0, 0.0, res, poses, obs_x, obs_y, capi.SIM_FOV, capi.SIM_MIN_ANGLE, 30.0)
    assert np.allclose(w_g, w_o, atol=2.0 / n)                       # goodness = paired fraction: at most a pair or two apart
    ok = (w_g > 0.5) & (w_o > 0.5)
    assert ok.sum() >= P // 2 and np.allclose(rel_g[ok], rel_o[ok], atol=2e-3)
    assert w_g[0] > 0.95 and np.abs(rel_g[0]).max() < 5e-3          # the true pose: everything pairs, no correction
    assert w_g[-1] < w_g[0]


def test_valid_poses_and_global_search(world):
    """CVPmap::isValidPose for a pose list against the oracle, and the reference's global localisation pipeline
    (valid candidates -> scorer -> topK -> best pose, localization.cpp:483-540) recovering a known pose."""
    import torch
    from laser_slam_amd import capi, localization
    import oracle_lib as ol
    ctx, fld, prob, res = world["ctx"], world["fld"], world["prob"], world["res"]
    sx, sy = prob.shape
    geom = (sx, sy, 0, 0, 0.0, 0.0, res)
    rng = np.random.default_rng(21)
    P = 4000
    poses = localization.random_poses(P, sx * res * 1.1, sy * res * 1.1, rng)        # some fall outside the map
    poses[:50, :2] -= 1.0                                                            # and some at negative coordinates
    d_prob = torch.from_numpy(prob).cuda(); d_poses = torch.from_numpy(poses).cuda()
    d_valid = torch.zeros(P, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    capi.valid_poses(fld, d_prob, d_poses, P, d_valid); ctx.sync()
    want = np.array([ol.valid_pose(prob, geom, float(p[0]), float(p[1])) for p in poses], np.uint8)
    assert np.array_equal(d_valid.cpu().numpy(), want)
    assert 0.2 < want.mean() < 0.95
    # global search with the beam model: the observation is the virtual scan of one valid candidate
    n = 181
    true_i = int(np.nonzero(want)[0][7])
    r = torch.zeros(1, n, device="cuda"); bad = torch.zeros(1, n, dtype=torch.uint8, device="cuda")
    capi.simulate_scans(fld, d_prob, d_poses[true_i:true_i + 1].contiguous(), 1, n, r, bad); ctx.sync()
    best, idx, sc = localization.global_search(fld, d_prob, d_poses, dict(r=r[0].contiguous(), bad=bad[0].contiguous()), scorer="beam", k=10)
    assert int(idx[0]) == true_i and np.array_equal(best, poses[true_i])
    assert np.all(np.diff(sc) <= 0) and want[idx].all()


def _fused_case(world, P, seed):
    torch = world["torch"]
    rng = np.random.default_rng(seed)
    n = 361
    ang = np.linspace(-np.pi / 2, np.pi / 2, n)
    d_x = torch.from_numpy((3.0 * np.cos(ang)).astype(np.float32)).cuda(); d_y = torch.from_numpy((3.0 * np.sin(ang)).astype(np.float32)).cuda()
    poses = np.stack([rng.uniform(1, 12, P), rng.uniform(1, 9, P), rng.uniform(-3, 3, P)], 1).astype(np.float32)
    nrm_p = rng.normal(0, 1, (P, 3)).astype(np.float32); nrm_r = rng.normal(0, 1, (P, 3)).astype(np.float32)
    return n, d_x, d_y, poses, nrm_p, nrm_r


@pytest.mark.parametrize("P", [1, 777, 20000, 300000, 2500000])
def test_fused_step_equals_call_by_call_step(world, P):
    """The fused step (lsa_pf_filter_*: four launches, one read-back) against the call-by-call entry points it replaces in
    the loop - lsa_pf_predict, lsa_score_poses, lsa_pose_argmax, lsa_pf_weights, the fixed-point resampling pieces,
    lsa_pf_roughen - bit for bit: predicted poses, scores, best particle, new particle set; N_eff sums to rounding."""
    torch, capi, ctx, fld = world["torch"], world["capi"], world["ctx"], world["fld"]
    from laser_slam_amd import pf
    n, d_x, d_y, poses, nrm_p, nrm_r = _fused_case(world, P, 50 + P % 7)
    mean, sigma, u0, K = [6.5, 5.0, 0.2], [2.0, 1.5, 1.0], 0.6180339, 0.2
    d_np, d_nr = torch.from_numpy(nrm_p).cuda(), torch.from_numpy(nrm_r).cuda()
    # call by call
    a = torch.from_numpy(poses).cuda(); torch.cuda.synchronize()
    capi.pf_predict(ctx, P, mean, sigma, d_np, a)
    prims = pf.DevicePrims(ctx, fld, d_x, d_y, n, 3)
    flt = pf.ShardedParticleFilter(P, prims, resample_threshold=2.0, device="cuda")
    s_a = prims.score(a); ctx.sync()
    new_a, info_a = flt.step(a, u0)
    capi.pf_roughen(ctx, P, K, d_nr, new_a); ctx.sync()
    # fused, phase by phase over torch (world 1) ...
    b = torch.from_numpy(poses).cuda(); torch.cuda.synchronize()
    fp = pf.FusedDevicePrims(ctx, fld, P, d_x, d_y, n, 3)
    ff = pf.FusedParticleFilter(P, fp, resample_threshold=2.0, roughen_K=K)
    new_b, info_b = ff.step(b, u0, predict=("predict", mean, sigma), normals_predict=d_np, normals_roughen=d_nr)
    ctx.sync()
    assert np.array_equal(b.cpu().numpy(), a.cpu().numpy())                          # predicted poses
    s_b = torch.as_tensor(pf._DeviceView(fp.flt.d_score, P, "<f4"), device="cuda")
    assert np.array_equal(s_b.cpu().numpy(), s_a.cpu().numpy())
    assert info_b["resampled"] and info_b["best_score"] == info_a["best_score"] and info_b["best_index"] == info_a["best_index"]
    assert np.array_equal(np.asarray(info_b["best_pose"], np.float32), a.cpu().numpy()[info_b["best_index"]])
    assert abs(info_b["weight_sum"] - info_a["weight_sum"]) < 1e-9 * info_a["weight_sum"] and abs(info_b["n_eff"] - info_a["n_eff"]) < 1e-9 * info_a["n_eff"]
    assert np.array_equal(new_b.cpu().numpy(), new_a.cpu().numpy())
    # ... and as the one call of the C ABI
    c = torch.from_numpy(poses).cuda(); c_out = torch.zeros(P, 3, device="cuda"); torch.cuda.synchronize()
    info_c = fp.flt.step(d_x, d_y, n, 3, c, c_out, P, P, u0, resample_threshold=2.0, roughen_K=K, d_normals_roughen=d_nr,
                         predict=("predict", mean, sigma), d_normals_predict=d_np)
    ctx.sync()
    assert info_c["resampled"] and info_c["best_index"] == info_a["best_index"] and info_c["n_eff"] == info_b["n_eff"]
    assert np.array_equal(c_out.cpu().numpy(), new_a.cpu().numpy())
    # below the threshold nothing is resampled and the output buffer stays untouched
    c_out.zero_(); torch.cuda.synchronize()
    info_d = fp.flt.step(d_x, d_y, n, 3, c, c_out, P, P, u0, resample_threshold=0.0)
    ctx.sync()
    assert not info_d["resampled"] and float(c_out.abs().sum()) == 0.0
    fp.close()


def test_fused_step_virtual_shards_equal_whole_set(world):
    """Three virtual shards on one GPU (keys and records concatenated by hand where the ranks would all-gather them): every
    shard's offspring range resolved from its own cumulative sums = the matching slice of the single-shard result, the
    motion-model prediction included."""
    torch, capi, ctx, fld = world["torch"], world["capi"], world["ctx"], world["fld"]
    from laser_slam_amd import pf
    P = 50001
    n, d_x, d_y, poses, nrm_p, nrm_r = _fused_case(world, P, 91)
    delta, sigma, u0 = [0.05, -0.02, 0.03], [0.05, 0.05, 0.02], 0.123
    d_np = torch.from_numpy(nrm_p).cuda()
    whole = torch.from_numpy(poses).cuda(); whole_out = torch.zeros(P, 3, device="cuda"); torch.cuda.synchronize()
    f0 = capi.PfFilter(fld, P)
    info0 = f0.step(d_x, d_y, n, 3, whole, whole_out, P, P, u0, resample_threshold=2.0, predict=("motion", delta, sigma), d_normals_predict=d_np)
    ctx.sync()
    th = whole.cpu().numpy()[:, 2]
    assert th.max() <= np.pi and th.min() > -np.pi
    world_n = 3
    parts, filts, keys, recs = [], [], [], []
    for r in range(world_n):
        first, cnt = pf.shard_bounds(P, r, world_n)
        part = torch.from_numpy(poses[first:first + cnt]).cuda(); torch.cuda.synchronize()
        f = capi.PfFilter(fld, cnt)
        f.score(d_x, d_y, n, 3, part, cnt, first, predict=("motion", delta, sigma), d_normals=d_np[first:first + cnt].contiguous())
        ctx.sync()
        keys.append(torch.as_tensor(pf._DeviceView(f.d_key, 1, "<i8"), device="cuda").clone())
        parts.append(part); filts.append(f)
    d_keys = torch.cat(keys); torch.cuda.synchronize()
    for r in range(world_n):
        filts[r].weights(parts[r], d_keys, world_n); ctx.sync()
        recs.append(torch.as_tensor(pf._DeviceView(filts[r].d_record, 6, "<i8"), device="cuda").clone())
    d_recs = torch.cat(recs); torch.cuda.synchronize()
    info, sums = filts[0].read(P, d_recs, world_n)
    assert info["best_index"] == info0["best_index"] and info["best_score"] == info0["best_score"]
    assert abs(info["n_eff"] - info0["n_eff"]) < 1e-9 * info0["n_eff"]
    assert np.array_equal(np.asarray(info["best_pose"], np.float32), np.asarray(info0["best_pose"], np.float32))
    W, U0, bounds = capi.pf_plan(sums, P, u0)
    got = []
    for r in range(world_n):
        lo, cnt = int(bounds[r]), int(bounds[r + 1] - bounds[r])
        out = torch.zeros(max(cnt, 1), 3, device="cuda"); torch.cuda.synchronize()
        filts[r].offspring(sum(sums[:r]), P, U0, W, lo, cnt, parts[r], out); ctx.sync()
        got.append(out.cpu().numpy()[:cnt])
    assert np.array_equal(np.concatenate(got), whole_out.cpu().numpy())
    for f in filts + [f0]:
        f.close()


def test_full_size_filter_step_properties(world):
    """BASELINE config 4's size: one particle-filter step over 1 000 000 particles (likelihood-field scores, best
    pose, weights and N_eff, the reference's standard resampler with one uniform per offspring, gather, estimate of
    the 8 best) checked through properties that do not need a second implementation at that size."""
    torch, capi, ctx, fld = world["torch"], world["capi"], world["ctx"], world["fld"]
    rng = np.random.default_rng(44)
    P, n = 1000000, 361
    ang = np.linspace(-np.pi / 2, np.pi / 2, n)
    d_x = torch.from_numpy((3.0 * np.cos(ang)).astype(np.float32)).cuda(); d_y = torch.from_numpy((3.0 * np.sin(ang)).astype(np.float32)).cuda()
    poses = np.stack([rng.uniform(1, 12, P), rng.uniform(1, 9, P), rng.uniform(-3, 3, P)], 1).astype(np.float32)
    d_p = torch.from_numpy(poses).cuda()
    d_s = torch.zeros(P, device="cuda"); torch.cuda.synchronize()
    fld.score_poses(d_x, d_y, n, 3, d_p, P, d_s); ctx.sync()
    sc = d_s.cpu().numpy()
    m, i = capi.pose_argmax(ctx, d_s, P)
    assert m == sc.max() and i == int(np.argmax(sc))                 # ties: the smaller index, like np.argmax
    d_w = torch.zeros(P, device="cuda"); torch.cuda.synchronize()
    s1, s2 = capi.pf_weights(ctx, d_s, P, m, d_w); ctx.sync()
    w = d_w.cpu().numpy()
    assert w.max() == 1.0 and abs(s1 - w.astype(np.float64).sum()) < 1e-9 * s1 and abs(s2 - (w.astype(np.float64) ** 2).sum()) < 1e-9 * s2
    u = rng.uniform(0, 1, P)
    d_par = torch.zeros(P, dtype=torch.int32, device="cuda"); torch.cuda.synchronize()
    capi.pf_resample_standard(ctx, d_w, P, torch.from_numpy(u).cuda(), P, d_par); ctx.sync()
    par = d_par.cpu().numpy()
    assert np.all(np.diff(par) >= 0) and par.min() >= 0 and par.max() < P
    cum = np.cumsum(w.astype(np.float64))
    want = np.searchsorted(cum, np.sort(u) * cum[-1], side="right")
    assert (par != want).mean() < 1e-3 and np.abs(par - want).max() <= 1
    assert w[par].min() > 0                                          # dead particles are never drawn
    d_new = torch.zeros(P, 3, device="cuda"); torch.cuda.synchronize()
    capi.gather_poses(ctx, d_p, d_par, P, d_new); ctx.sync()
    assert np.array_equal(d_new.cpu().numpy(), poses[par])
    mean, mw, md, idx = capi.pf_estimate(ctx, d_p, d_s, P, 8)
    order = np.lexsort((np.arange(P), -sc))[:8]                      # best first, equal scores in index order
    assert np.array_equal(idx, order)
    assert float(sc[par].mean()) > float(sc.mean())                  # the offspring sit on better poses


def test_topk_large_with_ties(world):
    """lsa_pose_topk on large score lists (radix select + candidate sort): distinct scores, a few ties at the
    k-th score, and more ties of the k-th score than the candidate buffer holds (falls back to the k passes) -
    always the k best, best first, equal scores in index order like threadLocalization::topK."""
    import torch
    from laser_slam_amd import capi
    ctx = world["ctx"]
    rng = np.random.default_rng(12)

    def check(sc, k):
        d = torch.from_numpy(sc).cuda(); torch.cuda.synchronize()
        idx, val = capi.pose_topk(ctx, d, len(sc), k)
        order = np.lexsort((np.arange(len(sc)), -sc.astype(np.float64)))[:k]
        assert np.array_equal(idx, order) and np.array_equal(val, sc[order])

    n = 300000
    check(rng.normal(0, 1, n).astype(np.float32), 60)
    check(rng.normal(0, 1, n).astype(np.float32), 1)
    check(-np.abs(rng.normal(0, 1, n)).astype(np.float32), 500)           # negative scores (log-likelihoods)
    check(np.round(rng.normal(0, 1, n), 2).astype(np.float32), 60)          # ties everywhere, a few hundred at the k-th score
    check(np.round(rng.normal(0, 1, n), 0).astype(np.float32), 60)          # tens of thousands of ties at the k-th score
    sc = np.zeros(n, np.float32); sc[rng.integers(0, n, 7)] = 1.0
    check(sc, 60)                                                            # 7 ones, then zeros in index order
