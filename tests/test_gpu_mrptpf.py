"""GPU: the MRPT-shaped facade (laser_slam_amd/host/mrptpf) behind src/pf_localization_main.cpp, SURVEY §8 row a24.

MRPT is absent, so there is nothing to pin the arithmetic to (parity unpinned): these tests check that the chain
runs on the GPU and localises - (1) an own driver making the calls of pf_localization_main.cpp in the same order,
(2) where it was built (oracle/_ref/pf_localization, compiled in the build container from the reference's file,
unchanged, against the facade) the reference's own main on a CARMEN log used as map and as run."""
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT

pytestmark = pytest.mark.gpu

LIBDIR = os.path.join(ROOT, "laser_slam_amd", "lib")
FACADE = os.path.join(ROOT, "laser_slam_amd", "host", "mrptpf")
LOG = os.path.join(GOLDEN, "fr079_head.log")


def laser_poses(path):
    out = []
    for line in open(path):
        t = line.split()
        if t and t[0] == "ROBOTLASER1":
            n = int(t[8]); m = int(t[9 + n])
            out.append([float(v) for v in t[10 + n + m:13 + n + m]])
    return np.array(out)


def test_facade_driver(tmp_path):
    exe = str(tmp_path / "mrpt_facade_test")
    subprocess.check_call(["g++", "-O1", "-std=c++11", "-I", FACADE, "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "mrpt_facade_test.cpp"), "-o", exe,
                           "-L", LIBDIR, "-llsa_host", "-llsa_core", "-Wl,-rpath," + LIBDIR])
    env = dict(os.environ, LSA_MRPT_SEED="7")
    res = subprocess.run([exe, LOG, str(tmp_path), "20000"], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout + res.stderr
    out = res.stdout
    assert "ini 2 100 200 hello 1 2.50 7" in out and "missing-throws 1" in out
    # (1, 2, 90 deg) (+) (0.5, 0, 10 deg) = (1, 2.5, 100 deg); (-) gives the increment back
    m = re.search(r"pose (\S+) (\S+) (\S+) \| (\S+) (\S+) (\S+)", out)
    got = [float(v) for v in m.groups()]
    assert np.allclose(got, [1.0, 2.5, np.deg2rad(100), 0.5, 0.0, np.deg2rad(10)], atol=1e-6)
    n_scans = len(laser_poses(LOG))
    m = re.search(r"map scans (\d+) cells (\d+) area (\S+) size (\d+) x (\d+)", out)
    assert int(m.group(1)) == n_scans and int(m.group(2)) > 2000 and float(m.group(3)) > 20.0
    assert re.search(r"gridmap round trip %s x %s, cells off by more than a grey level: 0" % (m.group(4), m.group(5)), out)
    m = re.search(r"pf steps (\d+) entries (\d+) worst-after-10 (\S+) last (\S+) min-ess (\S+) ess-now (\S+) size (\d+)", out)
    assert int(m.group(1)) == n_scans and int(m.group(2)) == 2 * n_scans and int(m.group(7)) == 20000
    # started within +-1.5 m / +-0.5 rad of the truth, the filter sits on the log's own poses once it has converged
    assert float(m.group(3)) < 0.35 and float(m.group(4)) < 0.25, out
    assert 0.0 < float(m.group(5)) <= 1.0 and 0.0 < float(m.group(6)) <= 1.0
    assert "scene 1 0" in out
    parts = np.loadtxt(tmp_path / "particles.txt")
    assert parts.shape == (20000, 4) and np.isfinite(parts).all()
    assert "GT" in open(tmp_path / "scene.3Dscene").read()


REF_MAIN = os.path.join(ROOT, "oracle", "_ref", "pf_localization")


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="oracle/_ref/pf_localization is built only where /root/reference exists")
def test_reference_pf_localization_main(tmp_path):
    """The reference's own main(), compiled unchanged, on fr079's head: map from the log (as '.simplemap'), run on the
    same log (as '.rawlog'), ground truth = the log's laser poses; its SUMMARY file and per-step outputs."""
    poses = laser_poses(LOG)
    simplemap = tmp_path / "fr079.simplemap"
    rawlog = tmp_path / "fr079.rawlog"
    for dst in (simplemap, rawlog):
        os.symlink(LOG, dst)
    gt = tmp_path / "gt.txt"
    with open(gt, "w") as f:                        # rawlog entry (two per laser line here), x, y, phi
        for k, p in enumerate(poses):
            f.write("%d %.6f %.6f %.6f\n" % (2 * (k + 1), p[0], p[1], p[2]))
    x0, y0, t0 = poses[0]
    ini = tmp_path / "pf.ini"
    ini.write_text("""
[LocalizationExperiment]
particles_count = 20000
logOutput_dir = %s/run
rawlog_file = %s
map_file = %s
ground_truth_path_file = %s
experimentRepetitions = 1
3DSceneFrequency = 30
SAVE_STATS_ONLY = false
init_PDF_mode = 0
init_PDF_min_x = %f
init_PDF_max_x = %f
init_PDF_min_y = %f
init_PDF_max_y = %f
init_PDF_min_phi_deg = %f
init_PDF_max_phi_deg = %f

[PF_options]
PF_algorithm = 0
resamplingMethod = 3
BETA = 0.5

[KLD_options]
KLD_maxSampleSize = 20000

[MetricMap]
occupancyGrid_count = 1

[MetricMap_occupancyGrid_00_creationOpts]
resolution = 0.10
""" % (tmp_path, rawlog, simplemap, gt, x0 - 1.5, x0 + 1.5, y0 - 1.5, y0 + 1.5, np.rad2deg(t0) - 25, np.rad2deg(t0) + 25))
    env = dict(os.environ, LSA_MRPT_SEED="11")
    res = subprocess.run([REF_MAIN, str(ini)], capture_output=True, text=True, timeout=900, env=env, cwd=tmp_path)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    run = tmp_path / "run_000"
    for name in ("gridmap.png", "gridmap_limits.txt", "cov_est.txt", "PF_stats.txt", "odo_est.txt"):
        assert (run / name).exists(), name
    stats = np.loadtxt(run / "PF_stats.txt")
    assert stats.shape == (len(poses), 3) and (stats[:, 0] == 20000).all() and (stats[1:, 1] > 0).all() and (stats[:, 1] <= 1.0 + 1e-9).all()
    assert (run / "particles" / "particles_030.txt").exists() and (run / "3D" / "progress_030.3Dscene").exists()
    # the application's own verdict: weighted localisation error against the ground truth, averaged over the run
    summary = [l for l in open(str(tmp_path / "run_SUMMARY.txt")) if not l.startswith("%")]
    vals = [float(v) for v in summary[-1].split()]
    assert int(vals[1]) == 20000 and vals[3] < 0.5, summary
    # and the last estimate it printed, against the log's last pose
    est = re.findall(r"PDF est: \(([-\d.]+),([-\d.]+),([-\d.]+)deg\)", res.stdout)
    ex, ey, _ = (float(v) for v in est[-1])
    assert np.hypot(ex - poses[-1][0], ey - poses[-1][1]) < 0.3
