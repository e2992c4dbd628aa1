"""CPU checks of the oracle's ray-cast simulator, beam model and particle-filter helpers
(oracle/sim_oracle.c) against first principles: these reference functions have no fixtures
(parity unpinned), so the restatement is pinned to geometry that can be worked out by hand."""
import numpy as np

import oracle_lib as ol

FOV, MIN_ANGLE = 4.712388981, 3 * -0.7853981635


def box_map(n=200, res=0.05):
    """Free square room with occupied walls 2 cells thick; map origin at the centre cell."""
    prob = np.full((n, n), 0.1, np.float32)
    prob[:2, :] = 0.9; prob[-2:, :] = 0.9; prob[:, :2] = 0.9; prob[:, -2:] = 0.9
    return prob, n // 2, n // 2, 0.0, 0.0, res


def test_simulated_ranges_match_room_geometry():
    prob, cx, cy, ox, oy, res = box_map()
    r, bad = ol.simulate_scan(prob, cx, cy, ox, oy, res, [0.0, 0.0, 0.0], 541, FOV, MIN_ANGLE, 30.0)
    assert not bad.any()
    # every ray ends on a wall: the hit point is within one step of the inner wall faces (+-4.9 m)
    AA = np.float32(FOV) / np.float32(540)
    ang = -np.float32(MIN_ANGLE) - np.arange(541) * float(AA)          # theta = -normAngle(0) = 0
    hx, hy = r * np.cos(ang), r * np.sin(ang)
    edge = np.maximum(np.abs(hx), np.abs(hy))
    assert np.all(edge > 4.9 - 2 * res) and np.all(edge < 4.9 + 2 * res)
    # ranges are whole numbers of steps
    assert np.allclose(r / res, np.round(r / res), atol=1e-3)


def test_simulator_flags_unknown_space_and_range_limit():
    prob, cx, cy, ox, oy, res = box_map()
    prob[150:, :] = 0.5                                                  # unknown band before the wall
    r, bad = ol.simulate_scan(prob, cx, cy, ox, oy, res, [0.0, 0.0, 0.0], 181, FOV, MIN_ANGLE, 30.0)
    assert bad.any() and not bad.all()
    r2, bad2 = ol.simulate_scan(box_map()[0], cx, cy, ox, oy, res, [0.0, 0.0, 0.0], 181, FOV, MIN_ANGLE, 2.0)
    assert bad2.all() and np.allclose(r2, 2.0, atol=res)                 # nothing within 2 m: max_ray_len steps, invalid


def test_beam_model_prefers_the_true_pose():
    prob, cx, cy, ox, oy, res = box_map()
    true = np.array([0.6, -0.4, 0.3], np.float32)
    obs_r, obs_bad = ol.simulate_scan(prob, cx, cy, ox, oy, res, true, 181, FOV, MIN_ANGLE, 30.0)
    poses = np.array([true, true + [0.3, 0, 0], true + [0, 0, 0.2], [-2, 2, 1.0]], np.float32)
    s = ol.beam_model(prob, cx, cy, ox, oy, res, poses, obs_r, obs_bad, FOV, MIN_ANGLE, 30.0)
    assert s[0] == np.float32(1 + 181) and s.argmax() == 0 and s[3] < s[1]


def test_topk_estimate_predict_roughen():
    rng = np.random.default_rng(5)
    score = rng.random(500).astype(np.float32)
    score[[7, 99, 300]] = score.max() + 1                                 # three equal maxima: index order
    idx = ol.topk(score, 8)
    assert list(idx[:3]) == [7, 99, 300]
    assert np.array_equal(np.sort(score)[::-1][:8], score[idx])
    poses = rng.normal(size=(500, 3)).astype(np.float32)
    mean, mw, md = ol.pf_estimate(poses, score, idx)
    w = score[idx].astype(np.float64); w /= w.sum()
    assert np.allclose(mean, (w[:, None] * poses[idx]).sum(0), atol=1e-12)
    assert np.isclose(mw, score[idx].astype(np.float64).sum() / 8)
    d = poses[idx][:, None, :2].astype(np.float64) - poses[idx][None, :, :2]
    assert np.isclose(md, (d ** 2).sum(-1).mean())
    nrm = rng.normal(size=(500, 3)).astype(np.float32)
    pp = ol.pf_predict(500, [1, 2, 0.5], [0.25, 0.25, 0.27], nrm)
    assert np.allclose(pp, np.array([1, 2, 0.5]) + nrm * [0.25, 0.25, 0.27], atol=1e-6)
    mm = ol.pf_motion(poses, [0.3, -0.1, 0.2], [0.05, 0.05, 0.02], nrm)
    inc = np.array([0.3, -0.1, 0.2]) + nrm * [0.05, 0.05, 0.02]
    c, s_ = np.cos(poses[:, 2].astype(np.float64)), np.sin(poses[:, 2].astype(np.float64))
    want = np.stack([poses[:, 0] + inc[:, 0] * c - inc[:, 1] * s_, poses[:, 1] + inc[:, 0] * s_ + inc[:, 1] * c,
                     np.arctan2(np.sin(poses[:, 2] + inc[:, 2]), np.cos(poses[:, 2] + inc[:, 2]))], 1)
    assert np.allclose(mm, want, atol=2e-6)
    rr = ol.pf_roughen(poses, 0.2, nrm)
    rootq = (poses.max(0).astype(np.float64) - poses.min(0)) * 0.2 * 500 ** (-1 / 3)
    assert np.allclose(rr, poses + nrm * rootq, atol=1e-6)


def test_standard_resampler_counts():
    """Standard_resampler (SIRFlt.cpp:39-115) + copy_resamples: a particle is drawn once per uniform inside its slice
    of the cumulative weights, whatever order the uniforms come in; the offspring are listed parent-ascending."""
    import oracle_lib as ol
    rng = np.random.default_rng(4)
    P = 500
    w = rng.uniform(0, 1, P).astype(np.float32) ** 4
    w[17:40] = 0.0                                       # dead particles are never drawn
    u = rng.uniform(0, 1, 800)
    parent, left = ol.standard_resample(w, u)
    assert left == 0 and np.all(np.diff(parent) >= 0)
    cum = np.cumsum(w.astype(np.float64))
    want = np.searchsorted(cum, np.sort(u) * cum[-1], side="right")
    assert (parent != want).sum() <= 1                   # (Kahan vs pairwise summation: a boundary case at most)
    assert not np.isin(parent, np.arange(17, 40)).any()
    # the order of the draws does not matter
    parent2, _ = ol.standard_resample(w, u[::-1].copy())
    assert np.array_equal(parent, parent2)
    # one heavy particle takes nearly everything
    w2 = np.full(P, 1e-6, np.float32); w2[123] = 1.0
    p3, _ = ol.standard_resample(w2, u)
    assert (p3 == 123).mean() > 0.99
