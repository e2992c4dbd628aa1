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


def _J(x, rho_ref, rho_cur, theta, corr):
    """sum_i (n_i^T (R p_i + t - p_j1))^2 over the given correspondences (the point-to-line objective)."""
    c, s = np.cos(x[2]), np.sin(x[2])
    i = np.nonzero(corr[:, 0] >= 0)[0]
    j1, j2 = corr[i, 0], corr[i, 1]
    p = np.stack([np.cos(theta[i]) * rho_cur[i], np.sin(theta[i]) * rho_cur[i]], 1)
    q1 = np.stack([np.cos(theta[j1]) * rho_ref[j1], np.sin(theta[j1]) * rho_ref[j1]], 1)
    q2 = np.stack([np.cos(theta[j2]) * rho_ref[j2], np.sin(theta[j2]) * rho_ref[j2]], 1)
    d = q2 - q1
    nrm = np.stack([-d[:, 1], d[:, 0]], 1) / np.linalg.norm(d, axis=1, keepdims=True)
    w = np.stack([c * p[:, 0] - s * p[:, 1] + x[0], s * p[:, 0] + c * p[:, 1] + x[1]], 1) - q1
    return float(np.sum(np.sum(nrm * w, 1) ** 2))


def test_plicp_covariance_against_finite_differences():
    """Censi's closed form, cov = sigma^2 H^-1 (A A^T) H^-1 with H = d2J/dx2 and A = d2J/dx dz over all readings z, against
    the same expression built from central differences of an independent numpy J (the oracle takes the projector's
    derivative by a 1 mm forward difference like sm_icp: agreement to a few 1e-3 relative)."""
    raw, _ = synth.make_scans(4)
    o = ol.Oracle(**ol.SYN1081)
    P = o.param_dict()
    n = 1081
    theta = (np.float32(P["fi_min"]) + np.float32(P["dfi"]) * np.arange(n, dtype=np.float32)).astype(np.float64)
    for k in (1,):
        out, cov, corr = ol.plicp_cov(raw[k - 1], raw[k], P["fi_min"], P["dfi"], P["min_range"], P["max_range"], 0.04)
        assert out[3] == 1.0 and (corr[:, 0] >= 0).sum() == out[6]
        assert np.allclose(cov, cov.T, rtol=1e-9, atol=1e-18) and np.all(np.linalg.eigvalsh(cov) > 0)
        x = out[:3].copy()
        rho_ref = raw[k - 1].astype(np.float64) / 100.0; rho_cur = raw[k].astype(np.float64) / 100.0
        hx = np.array([1e-4, 1e-4, 1e-5])

        def grad(rr, rc):
            g = np.zeros(3)
            for a in range(3):
                e = np.zeros(3); e[a] = hx[a]
                g[a] = (_J(x + e, rr, rc, theta, corr) - _J(x - e, rr, rc, theta, corr)) / (2 * hx[a])
            return g

        H = np.zeros((3, 3))
        for a in range(3):
            for b in range(3):
                ea = np.zeros(3); ea[a] = hx[a]; eb = np.zeros(3); eb[b] = hx[b]
                H[a, b] = (_J(x + ea + eb, rho_ref, rho_cur, theta, corr) - _J(x + ea - eb, rho_ref, rho_cur, theta, corr)
                           - _J(x - ea + eb, rho_ref, rho_cur, theta, corr) + _J(x - ea - eb, rho_ref, rho_cur, theta, corr)) / (4 * hx[a] * hx[b])
        used_ref = np.unique(corr[corr[:, 0] >= 0].ravel())
        used_cur = np.nonzero(corr[:, 0] >= 0)[0]
        S = np.zeros((3, 3))
        hz = 1e-4
        for which, idx in ((0, used_ref), (1, used_cur)):
            for j in idx:
                rp, rm = (rho_ref.copy(), rho_ref.copy()) if which == 0 else (rho_cur.copy(), rho_cur.copy())
                rp[j] += hz; rm[j] -= hz
                col = (grad(rp, rho_cur) - grad(rm, rho_cur)) / (2 * hz) if which == 0 else (grad(rho_ref, rp) - grad(rho_ref, rm)) / (2 * hz)
                S += np.outer(col, col)
        Hi = np.linalg.inv(H)
        cov_fd = 0.04 ** 2 * Hi @ S @ Hi
        assert np.allclose(cov, cov_fd, rtol=2e-2, atol=2e-2 * np.abs(cov_fd).max()), (cov, cov_fd)
        # a 4 cm reading noise on a room of this size: millimetres in position, a fraction of a milliradian in heading
        assert 1e-4 < np.sqrt(cov[0, 0]) < 2e-2 and 1e-5 < np.sqrt(cov[2, 2]) < 5e-3
