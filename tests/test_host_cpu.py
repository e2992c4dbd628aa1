"""CPU: the host-side logic of the C++ drop-in that needs no device - the text readers besides CARMEN and
pm_is_corridor - against fixtures made by the reference's own code compiled in place
(tests/golden/make_reader_golden.py).  The test program links liblsa_host.so / liblsa_core.so like a caller
would; constructing a CPolarMatch does not touch the GPU (the device context is created on the first match)."""
import os
import struct
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, SETS, golden_param, load, same_bits

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "laser_slam_amd", "lib")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    if not os.path.exists(os.path.join(LIBDIR, "liblsa_host.so")):
        pytest.skip("liblsa_host.so not built (python -c 'import __graft_entry__ as g; g.build()')")
    out = str(tmp_path_factory.mktemp("hostcpu") / "host_dropin_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "laser_slam_amd", "host"),
                           os.path.join(ROOT, "tests", "native", "host_dropin_test.cpp"), "-o", out,
                           "-L", LIBDIR, "-llsa_host", "-llsa_core", "-Wl,-rpath," + LIBDIR])
    return out


def test_text_readers_vs_reference_fixtures(exe, tmp_path):
    """CPolarMatch::readRawSeed, readSICK and construct511Data of the drop-in on synthetic files, against what the
    reference's own readers made of the same files."""
    z = np.load(os.path.join(GOLDEN, "readers.npz"))
    for kind, laser, fn, key in (("rawseed", "LMS211", "readers_rawseed.csv", "rawseed"), ("sick", "LMS511", "readers_sick.txt", "sick")):
        out = str(tmp_path / (kind + ".bin"))
        res = subprocess.run([exe, "readers", kind, laser, os.path.join(GOLDEN, fn), out], capture_output=True, text=True, timeout=300)
        assert res.returncode == 0, res.stderr[-2000:]
        raw = open(out, "rb").read()
        S = struct.unpack_from("<i", raw, 0)[0]
        n = z[key + "_r"].shape[1]
        assert S == z[key + "_r"].shape[0]
        rec = np.frombuffer(raw, np.float32, S * 4 * n, 4).reshape(S, 4, n)
        assert same_bits(rec[:, 0], z[key + "_r"]) and same_bits(rec[:, 1], z[key + "_x"]) and same_bits(rec[:, 2], z[key + "_y"])
        assert np.array_equal(rec[:, 3].view(np.int32), z[key + "_bad"])
    out = str(tmp_path / "t5.bin")
    res = subprocess.run([exe, "readers", "511", "LMS511", os.path.join(GOLDEN, "readers_511.txt"), out], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    raw = open(out, "rb").read()
    S = struct.unpack_from("<i", raw, 0)[0]
    n = z["t511_r"].shape[1]
    assert S == len(z["t511_ok"])
    rec = np.frombuffer(raw, np.int32, S * (1 + 4 * n), 4).reshape(S, 1 + 4 * n)
    assert np.array_equal(rec[:, 0], z["t511_ok"])
    ok = z["t511_ok"] == 1
    body = rec[:, 1:].reshape(S, 4, n)
    assert np.array_equal(body[ok, 0], z["t511_r"][ok].view(np.int32)) and np.array_equal(body[ok, 1], z["t511_x"][ok].view(np.int32))
    assert np.array_equal(body[ok, 2], z["t511_y"][ok].view(np.int32)) and np.array_equal(body[ok, 3], z["t511_bad"][ok])


def _corridor(exe, tmp_path, laser, fi, r, bad, seg):
    fin = str(tmp_path / "cor.bin")
    with open(fin, "wb") as f:
        f.write(struct.pack("<i", len(r)))
        for k in range(len(r)):
            f.write(np.ascontiguousarray(r[k], np.float32).tobytes()); f.write(np.ascontiguousarray(bad[k], np.int32).tobytes())
            f.write(np.ascontiguousarray(seg[k], np.int32).tobytes())
    res = subprocess.run([exe, "corridor", laser] + fi + [fin], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    return np.array([int(t) for t in res.stdout.split()], np.int32)


@pytest.mark.parametrize("name", ["mit", "fr079", "intel", "indoor"])
def test_pm_is_corridor_on_log_scans(name, exe, tmp_path):
    """pm_is_corridor (ZHPolar_Match.cpp:2058-2186) on the preprocessed scans of the logs' heads plus two degenerate
    segmentations (it throws 3), against the reference's answers."""
    z = np.load(os.path.join(GOLDEN, "readers.npz"))
    laser = SETS[name][0]
    g = load(name); gp = golden_param(g)
    n = g["pre_r"].shape[1]
    zero = np.zeros(n, np.int32); one = zero.copy(); one[10:12] = 1
    r = list(g["pre_r"]) + [g["pre_r"][0], g["pre_r"][0]]
    bad = list(g["pre_bad"]) + [zero, zero]
    seg = list(g["pre_seg"]) + [zero, one]
    fi = [repr(float(np.float32(gp["fi_min"]))), repr(float(np.float32(gp["fi_max"]))), str(int(gp["max_range"]))]
    got = _corridor(exe, tmp_path, laser, fi, r, bad, seg)
    assert np.array_equal(got, z["corridor_" + name])


def test_pm_is_corridor_synthetic(exe, tmp_path):
    z = np.load(os.path.join(GOLDEN, "readers.npz"))
    got = _corridor(exe, tmp_path, "LMS511", ["-", "-", "-"], z["corridor_syn_r"], z["corridor_syn_bad"], z["corridor_syn_seg"])
    assert np.array_equal(got, z["corridor_syn"]) and set(got.tolist()) == {0, 1}
