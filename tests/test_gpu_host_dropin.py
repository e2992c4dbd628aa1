"""GPU: the C++ drop-in classes (CPolarMatch, CPMap, CCanonicalMatcher, OrientedPoint2D)
driven like the reference's own callers, compared bit-for-bit with the golden vectors."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, SETS, load, same_bits

pytestmark = pytest.mark.gpu

LIBDIR = os.path.join(ROOT, "laser_slam_amd", "lib")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("dropin") / "host_dropin_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "laser_slam_amd", "host"),
                           os.path.join(ROOT, "tests", "native", "host_dropin_test.cpp"), "-o", out,
                           "-L", LIBDIR, "-llsa_host", "-llsa_core", "-Wl,-rpath," + LIBDIR])
    return out


def hexf(tok):
    return np.float32(float.fromhex(tok))


@pytest.mark.parametrize("name", ["mit", "intel", "indoor"])
def test_cpolarmatch_dropin(name, exe):
    g = load(name)
    laser, fn, tag = SETS[name]
    n_pairs = 12
    res = subprocess.run([exe, "match", laser, os.path.join(GOLDEN, fn), tag, str(n_pairs)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    rows = [l.split() for l in res.stdout.splitlines()]
    by = lambda k: [r for r in rows if r and r[0] == k]
    for r in by("fmatch"):
        b = int(r[1])
        got = np.array([hexf(t) for t in r[2:5]], np.float32)
        assert same_bits(got, g["fmatch_pose"][b]), (b, got, g["fmatch_pose"][b])
        pair = np.array([hexf(t) for t in r[5:7]], np.float32)
        assert same_bits(pair, g["fmatch_pair"][b])
    assert len(by("fmatch")) == n_pairs
    for nm in ("psm", "icp"):
        for r in by(nm):
            b, status = int(r[1]), int(r[2])
            assert (status != 0) == (g[nm + "_status"][b] != 0)
            if status == 0:
                assert same_bits(np.array([hexf(t) for t in r[3:6]], np.float32), g[nm + "_pose"][b])
                assert same_bits(hexf(r[6]), g[nm + "_ret"][b])
                assert int(r[7]) == g[nm + "_nassoc"][b]
                if int(r[7]) > 0:
                    assert same_bits(np.array([hexf(r[8]), hexf(r[9])], np.float32), g[nm + "_exy"][b])
            else:
                assert status == 1                      # the reference throws 1
                assert all(hexf(t) == 0 for t in r[3:6])   # lsa untouched
    for r in by("batch"):
        b = int(r[1])
        assert (int(r[2]) != 0) == (g["icp_status"][b] != 0)
        assert same_bits(np.array([hexf(t) for t in r[3:6]], np.float32), g["icp_pose"][b])
    # stage methods against the CPU oracle
    from oracle_lib import Oracle
    o = Oracle(laser)
    from helpers import golden_param
    gp = golden_param(g)
    o.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
    act = (g["pre_r"][1], g["pre_bad"][1], g["pre_seg"][1])
    nr, nb = o.scan_project(*act, np.array([3.5, -2.25, 0.03], np.float32))
    h = 2166136261
    for a, b_ in zip(nr.view(np.uint32).tolist(), nb.tolist()):
        h = ((h ^ a) * 16777619) & 0xffffffff
        h = ((h ^ (b_ & 0xffffffff)) * 16777619) & 0xffffffff
    st = by("stage")[0]
    assert int(st[1], 16) == h
    so, dth = o.orientation_search(g["pre_r"][0], g["pre_bad"][0], nr, nb)
    assert (int(st[2]) != 0) == (so != 0)
    if so == 0:
        assert hexf(st[3]) == np.float32(dth)
    stt, dx, dy, err = o.translation_estimation(g["pre_r"][0], g["pre_bad"][0], nr, nb)
    assert (int(st[4]) != 0) == (stt != 0)
    if stt == 0:
        assert same_bits(np.array([hexf(t) for t in st[5:8]], np.float32), np.array([dx, dy, err], np.float32))
    assert int(by("corridor")[0][1]) in (0, 1, -2, -3)
    csm = by("csm")[0]
    assert float(csm[1]) == 2.0 and float(csm[2]) in (0.0, -1.0)          # first frame stored, then a match
    if float(csm[2]) == 0.0:
        # CCanonicalMatcher: median filter, then the point-to-line ICP with the reference's options
        # (parity unpinned: against oracle/plicp_oracle.c), stored in cm / rad
        from oracle_lib import Oracle, plicp
        o = Oracle(laser)
        r_all, _, _, bad_all, _ = o.read_carmen(os.path.join(GOLDEN, fn), tag)
        gp = golden_param(g)
        o.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
        ra, rb = o.median(r_all[0]), o.median(r_all[1])
        P = o.param_dict()
        want = plicp(ra, rb, P["fi_min"], P["dfi"], P["min_range"], P["max_range"])
        got = np.array([hexf(t) for t in csm[3:6]], np.float32)
        assert want[3] == 1.0
        assert np.allclose(got, [want[0] * 100, want[1] * 100, want[2]], rtol=0, atol=2e-4)
    se2 = [float.fromhex(t) for t in by("se2")[0][1:]]
    assert np.allclose(se2[3:], [0.25, -0.5, 0.125], atol=1e-12)          # p (-) (p (+) q) == q


@pytest.mark.parametrize("name", ["mit"])
def test_cpmap_dropin(name, exe, tmp_path):
    z = np.load(os.path.join(GOLDEN, "pmap_%s.npz" % name))
    G = z["geom"]
    S, nb = z["gx"].shape
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<4i2fd2i", int(G[0]), int(G[1]), int(G[2]), int(G[3]), float(G[4]), float(G[5]), float(G[6]), S, nb))
        f.write(z["gx"].astype(np.float32).tobytes()); f.write(z["gy"].astype(np.float32).tobytes())
        f.write(z["robot"].astype(np.float32).tobytes())
    res = subprocess.run([exe, "pmap", fin, fout], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    raw = open(fout, "rb").read()
    cells = int(G[0]) * int(G[1])
    lay = np.frombuffer(raw, np.float32, 3 * cells).reshape(3, int(G[0]), int(G[1]))
    assert np.array_equal(lay[0], z["hit"].astype(np.float32))
    assert np.array_equal(lay[1], z["sum"].astype(np.float32))
    assert same_bits(lay[2], z["prob"])
    nc = struct.unpack_from("<i", raw, 12 * cells)[0]
    cx = np.frombuffer(raw, np.float32, nc, 12 * cells + 4)
    cy = np.frombuffer(raw, np.float32, nc, 12 * cells + 4 + 4 * nc)
    assert same_bits(cx, z["cloud_x"]) and same_bits(cy, z["cloud_y"])


def test_cpmap_dropin_queue(exe, tmp_path):
    """updateMap() queues scans and flushes them in batches of 256 (or when a row wider than the queue's arrives, or
    when a reader needs the counts): 20 passes over the golden scans - 600 per-scan calls with ragged lengths - must
    give exactly 20 x the golden counts and, the ratio being the same, the golden probabilities."""
    z = np.load(os.path.join(GOLDEN, "pmap_mit.npz"))
    G = z["geom"]
    reps = 20
    gx = np.tile(z["gx"].astype(np.float32), (reps, 1)); gy = np.tile(z["gy"].astype(np.float32), (reps, 1))
    rob = np.tile(z["robot"].astype(np.float32), (reps, 1))
    assert len(set(rob[:, 2].tolist())) > 1                       # ragged: the queue's row width has to grow on the way
    S, nb = gx.shape
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<4i2fd2i", int(G[0]), int(G[1]), int(G[2]), int(G[3]), float(G[4]), float(G[5]), float(G[6]), S, nb))
        f.write(gx.tobytes()); f.write(gy.tobytes()); f.write(rob.tobytes())
    res = subprocess.run([exe, "pmap", fin, fout], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    raw = open(fout, "rb").read()
    cells = int(G[0]) * int(G[1])
    lay = np.frombuffer(raw, np.float32, 3 * cells).reshape(3, int(G[0]), int(G[1]))
    assert np.array_equal(lay[0], reps * z["hit"].astype(np.float32))
    assert np.array_equal(lay[1], reps * z["sum"].astype(np.float32))
    assert same_bits(lay[2], z["prob"])
    print(res.stderr.strip().splitlines()[-1])               # the per-scan signature's rate, host buffers in


def test_offline_driver_runlog_improved(exe, tmp_path):
    """CPolarMatch::runOurFile -> runlogImproved (ZHPolar_Match.cpp:353-360, 736-852) on the in-house
    LMS151 log: the trajectory file against the same driver logic replayed with the CPU oracle."""
    from oracle_lib import Oracle
    laser, fn, tag = SETS["indoor"]
    n = 20
    out = str(tmp_path / "traj.log")
    res = subprocess.run([exe, "runlog", "our", laser, os.path.join(GOLDEN, fn), out, str(n)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.loadtxt(out).reshape(-1, 3)
    o = Oracle(laser)
    r_all, _, _, bad_all, _ = o.read_carmen(os.path.join(GOLDEN, fn), tag)
    PI = 3.141592654

    def norm_a(a):
        m = int(a / (2.0 * PI)); a = a - m * PI
        if a < -PI: a += 2.0 * PI
        if a >= PI: a -= 2.0 * PI
        return a

    rx = ry = th = 0.0
    last = (0.0, 0.0, 0.0)
    prev = ref = None
    want = []
    for k in range(min(n, len(r_all))):
        ls = o.preprocess(r_all[k], bad_all[k])
        if prev is None:
            prev = ref = ls
            continue
        st, pose, _, _ = o.match(0, ref, ls)
        failed = st != 0
        if failed:
            pose = np.zeros(3, np.float32)
        exy, _ = o.error_index2(prev, ls, (0, 0, 0), pose)
        if failed or np.sqrt(np.float32(exy[0] + exy[1])) > np.float32(0.05):
            ref = prev
            st, pose, _, _ = o.match(0, ref, ls)
            if st != 0:
                st, pose, _, _ = o.match(1, ref, ls)
                if st != 0:
                    continue
            rx, ry, th = last
            ref = prev
        xx = float(pose[0]) * np.cos(th) - float(pose[1]) * np.sin(th) + rx
        yy = float(pose[0]) * np.sin(th) + float(pose[1]) * np.cos(th) + ry
        tth = norm_a(th + float(pose[2]))
        want.append((xx, yy, tth))
        prev = ls
        last = (xx, yy, tth)
    want = np.array(want)
    assert got.shape == want.shape and len(want) >= n - 3
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5)          # the file holds 6 significant digits


def test_offline_driver_runlog_flirt(exe, tmp_path):
    """CPolarMatch::runFlirtFile -> runlog (ZHPolar_Match.cpp:332-341, 449-647) on the head of mit-cscail.log: the
    drop-in batches the frames' matches (pm_psm, pm_icp from its result, pm_error_index2) and keeps only the
    bookkeeping sequential; the trajectory file against the driver's logic replayed frame by frame with the oracle."""
    from oracle_lib import Oracle
    laser, fn, tag = SETS["mit"]
    n = 40
    out = str(tmp_path / "traj.log")
    res = subprocess.run([exe, "runlog", "flirt", laser, os.path.join(GOLDEN, fn), out, str(n)],
                         capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.loadtxt(out).reshape(-1, 3)
    o = Oracle(laser)
    r_all, _, _, bad_all, pose_all = o.read_carmen(os.path.join(GOLDEN, fn), tag)

    def ominus(a, b):       # a.ominus(b): b in a's frame
        c, s = np.cos(a[2]), np.sin(a[2])
        return np.array([(b[0] - a[0]) * c + (b[1] - a[1]) * s, -(b[0] - a[0]) * s + (b[1] - a[1]) * c, b[2] - a[2]])

    def oplus(a, b):
        c, s = np.cos(a[2]), np.sin(a[2])
        return np.array([a[0] + b[0] * c - b[1] * s, a[1] + b[0] * s + b[1] * c, a[2] + b[2]])

    want = []
    prev = None
    rx = ry = th = 0.0
    last_odo = None
    bridged = 0
    for k in range(min(n, len(r_all))):
        ls = o.preprocess(r_all[k], bad_all[k])
        odo = pose_all[k].astype(np.float64)                         # PMScan keeps the log's pose as float
        if prev is None:
            last_odo = odo; rx, ry, th = odo[0] * 100, odo[1] * 100, odo[2]
            want.append((odo[0], odo[1], th)); prev = ls
            continue
        big_turn = abs(odo[2] - th) > 0.6982
        bridge = big_turn
        if not bridge:
            st, p_psm, _, _ = o.match(0, prev, ls)
            init = p_psm if st == 0 else np.zeros(3, np.float32)
            st2, p_icp, _, _ = o.match(1, prev, ls, (0, 0, 0), init)
            failed = st2 != 0
            rel = p_icp if st2 == 0 else init
            exy, _ = o.error_index2(prev, ls, (0, 0, 0), rel)
            err_idx = np.sqrt(np.float32(exy[0] + exy[1]))
            xx = float(rel[0]) * np.cos(th) - float(rel[1]) * np.sin(th) + rx
            yy = float(rel[0]) * np.sin(th) + float(rel[1]) * np.cos(th) + ry
            tth = th + float(rel[2])
            bridge = bool(err_idx > np.float32(10)) or (xx - rx) ** 2 + (yy - ry) ** 2 > 10000 or failed
        if bridge:
            bridged += 1
            cur = oplus(np.array([rx / 100.0, ry / 100.0, th]), ominus(last_odo, odo))
            rx, ry, th = cur[0] * 100, cur[1] * 100, cur[2]
            want.append((cur[0], cur[1], cur[2]) if big_turn else (rx / 100.0, ry / 100.0, th))
        else:
            rx, ry, th = xx, yy, tth
            want.append((xx / 100.0, yy / 100.0, tth))
        prev = ls; last_odo = odo
    want = np.array(want)
    assert got.shape == want.shape == (min(n, len(r_all)), 3)
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5)          # the file holds 6 significant digits
    assert bridged < len(want) // 2                                # most frames are matched, not bridged


def test_accuracy_harness(exe, tmp_path):
    """CPolarMatch::testAccuracy (ZHPolar_Match.cpp:364-447) on the head of mit-cscail.log: the two error logs against
    the harness's bookkeeping replayed with the golden poses of pm_psm / pm_icp (reference-made, zero prior)."""
    from oracle_lib import Oracle
    name = "mit"
    g = load(name)
    laser, fn, tag = SETS[name]
    fp, fi = str(tmp_path / "err_psm.log"), str(tmp_path / "err_icp.log")
    res = subprocess.run([exe, "accuracy", laser, os.path.join(GOLDEN, fn), fp, fi], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    failed = [l.split() for l in res.stdout.splitlines() if l.startswith("failed")][0]
    o = Oracle(laser)
    _, _, _, _, pose_all = o.read_carmen(os.path.join(GOLDEN, fn), tag)
    S = len(pose_all)
    assert int(failed[1]) == int((g["psm_status"][:S - 1] != 0).sum()) and int(failed[2]) == int((g["icp_status"][:S - 1] != 0).sum())
    for which, path in (("psm", fp), ("icp", fi)):
        want = []
        for b in range(S - 1):
            a, c = pose_all[b].astype(np.float64), pose_all[b + 1].astype(np.float64)
            cs, sn = np.cos(a[2]), np.sin(a[2])
            gt = np.array([(c[0] - a[0]) * cs + (c[1] - a[1]) * sn, -(c[0] - a[0]) * sn + (c[1] - a[1]) * cs, c[2] - a[2]])
            p = g[which + "_pose"][b].astype(np.float64) if g[which + "_status"][b] == 0 else np.zeros(3)
            dis = np.hypot(p[0] / 100.0 - gt[0], p[1] / 100.0 - gt[1]); ang = abs(p[2] - gt[2])
            if dis < 1 and ang < 0.6982:
                want.append((dis, ang))
        got = np.loadtxt(path).reshape(-1, 2)
        want = np.array(want)
        assert got.shape == want.shape and len(want) > (S - 1) // 2
        assert np.allclose(got, want, rtol=2e-5, atol=1e-9)          # the file holds 6 significant digits


def test_run511_driver(exe, tmp_path):
    """CPolarMatch::run511Data (ZHPolar_Match.cpp:678-734, what the reference's own src/zhpsm/test.cpp runs): a file of
    "timestamp n r_1 .. r_n" lines made from the head of fr079.log, chained pm_psm against the oracle's replay."""
    from oracle_lib import Oracle
    laser, fn, tag = SETS["fr079"]
    o = Oracle(laser)
    r_all, _, _, _, _ = o.read_carmen(os.path.join(GOLDEN, fn), tag)
    n = 25
    src = tmp_path / "lms511.txt"
    with open(src, "w") as f:
        for k in range(n):
            f.write("%d.5 %d %s\n" % (1000 + k, r_all.shape[1], " ".join(repr(float(np.float32(v)) / 100.0) for v in r_all[k])))
        f.write("7 12 1.0 2.0\n")                                    # a record of the wrong size is skipped
    out = str(tmp_path / "traj511.log")
    res = subprocess.run([exe, "run511", laser, str(src), out, "-1"], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    got = np.loadtxt(out).reshape(-1, 3)
    o = Oracle(laser)                 # run511Data reads no CARMEN header: the preset's bearings (0 .. 180 deg) stay
    P = o.param_dict()
    rx = ry = th = 0.0
    want = [(0.0, 0.0, 0.0)]
    last = None
    for k in range(n):
        r = (np.array([float(np.float32(v)) / 100.0 for v in r_all[k]]) * 100.).astype(np.float32)
        r = np.where(r < P["min_range"], np.float32(P["max_range"] + 1), r).astype(np.float32)
        ls = o.preprocess(r)
        if last is None:
            last = ls
            continue
        st, pose, _, _ = o.match(0, last, ls)
        if st != 0:
            continue
        xx = float(pose[0]) * np.cos(th) - float(pose[1]) * np.sin(th) + rx
        yy = float(pose[0]) * np.sin(th) + float(pose[1]) * np.cos(th) + ry
        rx, ry, th = xx, yy, th + float(pose[2])
        last = ls
        want.append((rx, ry, th))
    want = np.array(want)
    assert got.shape == want.shape and len(want) >= n - 3
    assert np.allclose(got, want, rtol=2e-5, atol=2e-5)


def test_ccanonicalmatcher_ldp_and_covariance(exe):
    """laser_data-typed FMatch == PMScan-typed FMatch; FMatchKeyFrame2's covariance == the oracle's closed form on the
    median-filtered frames (parity unpinned: CSM absent); setThreshold(dist, angle) in the reference's argument order."""
    from oracle_lib import Oracle, plicp_cov
    from helpers import golden_param
    laser, fn, tag = SETS["mit"]
    res = subprocess.run([exe, "csmx", laser, os.path.join(GOLDEN, fn), tag], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-2000:]
    rows = {r.split()[0] + (r.split()[1] if r.startswith("kf") else ""): r.split() for r in res.stdout.splitlines() if r.strip()}
    pm, ld = rows["pm"], rows["ld"]
    assert float(pm[1]) == 0.0 and float(ld[1]) == 0.0 and int(ld[6]) == 1
    assert np.array_equal([hexf(pm[2]), hexf(pm[3]), hexf(pm[4])],
                          [np.float32(float.fromhex(ld[2]) * 100.), np.float32(float.fromhex(ld[3]) * 100.), np.float32(float.fromhex(ld[4]))])
    g = load("mit")
    o = Oracle(laser)
    gp = golden_param(g)
    r_all, _, _, _, _ = o.read_carmen(os.path.join(GOLDEN, fn), tag)
    o.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
    P = o.param_dict()
    med = [o.median(r_all[k]) for k in range(4)]
    assert float(rows["kf0"][2]) == 2.0                              # first frame stored
    for k in range(1, 4):
        row = rows["kf%d" % k]
        want, wcov, _ = plicp_cov(med[k - 1], med[k], P["fi_min"], P["dfi"], P["min_range"], P["max_range"], 0.04)
        assert float(row[2]) == 0.0 and want[3] == 1.0
        got = np.array([float.fromhex(t) for t in row[3:6]])
        assert np.allclose(got, [want[0] * 100, want[1] * 100, want[2]], rtol=0, atol=2e-4)
        cov = np.array([float.fromhex(t) for t in row[6:15]]).reshape(3, 3)
        assert np.allclose(cov, wcov, rtol=1e-5, atol=1e-9 * np.abs(wcov).max()), (cov, wcov)
    thr = rows["thr"]
    assert [int(t) for t in thr[1:4]] == [0, 1, 1] and abs(float(thr[4]) - 0.25) < 1e-6 and abs(float(thr[5]) - 0.2) < 1e-6
