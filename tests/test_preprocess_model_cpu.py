"""CPU: the two reformulations behind preprocess_scan_kernel (laser_slam_amd/csrc/match_kernels.hip), restated in numpy
and held to the oracle's sequential pm_median_filter / pm_segment_scan (ZHPolar_Match.cpp:1610-1639, 1495-1576; the oracle
itself is pinned to the reference's goldens in test_oracle_golden.py):
  * the in-place median filter is the unique fixed point of block-wise sweeps (every block walked from the current values
    of the two beams in front of it);
  * the segmentation's carried state reduces to three cases, a beam's data fix a map of those cases, and the maps
    compose (blocks composed separately and chained give the sequential walk).
Random scans with spikes, steps, far readings and flagged beams, and patterns built to carry a disturbance far."""
import numpy as np
import pytest

from oracle_lib import SYN1081, Oracle

f32 = np.float32
LIM = f32(20.0)          # LSA_SEG_MAX_DIST


def med5(w):
    w = list(w)
    for hi in range(4, 0, -1):
        for k in range(hi):
            if w[k] > w[k + 1]:
                w[k], w[k + 1] = w[k + 1], w[k]
    return w[2]


def median_by_sweeps(raw, lanes=64):
    n = len(raw)
    per = -(-n // lanes)
    f = raw.copy()
    sweeps = 0
    while True:
        sweeps += 1
        start = f.copy()                       # what the lanes read of the blocks in front of them
        changed = False
        for b0 in range(0, n, per):
            f2 = start[b0 - 2] if b0 >= 2 else f32(0)
            f1 = start[b0 - 1] if b0 >= 1 else f32(0)
            for i in range(b0, min(b0 + per, n)):
                a0 = raw[i]
                w3 = raw[i + 1] if i + 1 <= n - 1 else a0
                w4 = raw[i + 2] if i + 2 <= n - 1 else w3
                m = med5([f2 if i >= 2 else (f1 if i == 1 else a0), f1 if i >= 1 else a0, a0, w3, w4])
                if m.tobytes() != f[i].tobytes():
                    f[i] = m; changed = True
                f2, f1 = f1, m
        if not changed:
            return f, sweeps
        assert sweeps <= lanes + 2


A, B, D = 0, 1, 2


def segment_by_maps(r, bad, lanes=64):
    n = len(r)
    maps = np.zeros((n, 3), np.int64); cont = np.zeros(n, bool)
    maps[:] = [A, B, D]
    for i in range(2, n):
        bi = bad[i] != 0
        c1 = abs(f32(r[i] - r[i - 1])) < LIM
        ex = f32(np.float64(r[i]) - (2.0 * np.float64(r[i - 1]) - np.float64(r[i - 2])))
        c2 = abs(ex) < LIM
        nb = (not bi) and bad[i - 1] == 0 and bad[i - 2] == 0
        cont[i] = (not bi) and c1
        maps[i] = [A if ((not bi) and (c1 or c2)) else D, A if cont[i] else (A if (nb and c2) else B), A if cont[i] else B]
    per = -(-n // lanes)
    # a block's maps composed; the blocks chained
    state0 = A if abs(f32(r[0] - r[1])) < LIM else B
    entry, st = [], state0
    for b0 in range(0, n, per):
        entry.append(st)
        tot = np.array([A, B, D])
        for i in range(b0, min(b0 + per, n)):
            tot = maps[i][tot]
        st = tot[st]
    flags = np.zeros(n + 2, np.int64); ids = np.zeros(n, np.int64)
    adv_total = 0
    for k, b0 in enumerate(range(0, n, per)):
        st = entry[k]
        for i in range(b0, min(b0 + per, n)):
            if i >= 2:
                sa = maps[i][st]
                if st == A and sa == D: adv_total += 1
                if st != A and sa == B: flags[i] = 1
                if st == B and sa == A and not cont[i]: flags[i] = 2
                st = sa
            elif i == 1 and state0 == B:
                flags[i] = 1
            ids[i] = 1 + adv_total
    seg = np.array([0 if ((flags[j + 1] & 1) and not (flags[j + 2] & 2)) else ids[j] for j in range(n)], np.int32)
    return seg


def scans(n, count, seed):
    rng = np.random.default_rng(seed)
    for k in range(count):
        base = 400.0 + 250.0 * np.sin(np.arange(n) * rng.uniform(0.01, 0.2)) + np.where(np.arange(n) > rng.integers(0, n), rng.uniform(-300, 300), 0.0)
        r = (base + rng.normal(0, rng.choice([0.5, 3.0, 15.0]), n)).astype(f32)
        r += (rng.normal(0, 60.0, n) * (rng.random(n) < 0.06)).astype(f32)
        r[rng.random(n) < 0.01] = f32(9000.0)
        if k % 4 == 1:
            r = np.where(np.arange(n) % 2 == 0, r, r[::-1]).astype(f32)           # jagged: disturbances carry
        if k % 4 == 2:
            r = np.round(r / 25.0).astype(f32) * f32(25.0)                        # plateaus and exact ties at the 20 cm limit's scale
        bad = (rng.random(n) < rng.choice([0.0, 0.03, 0.2])).astype(np.int32)
        yield r, bad


@pytest.mark.parametrize("points", [1081, 361, 67, 8])
def test_median_fixed_point_and_segmentation_maps(points):
    par = dict(SYN1081, points=points, window=min(SYN1081["window"], (points - 1) // 2))
    o = Oracle(**par)
    worst = 0
    for r, bad in scans(points, 40 if points > 400 else 80, points):
        want_r = o.median(r)
        got_r, sweeps = median_by_sweeps(r)
        worst = max(worst, sweeps)
        assert got_r.tobytes() == want_r.tobytes()
        fr, fbad, fseg = o.preprocess(r, bad)
        assert np.array_equal(segment_by_maps(fr, fbad), fseg)
    assert worst <= 66
