"""CPU: the C-ABI library loads, exports every symbol include/lsa_core.h declares, and
its host-only entry points (parameters, bearing tables, CARMEN reader) reproduce the
golden vectors.  No compute calls (those need a GPU and fail loudly without one)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from helpers import GOLDEN, ROOT, SETS, golden_param, load, same_bits


def declared_functions():
    src = open(os.path.join(ROOT, "include", "lsa_core.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lsa_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from laser_slam_amd import capi
    L = capi.lib()
    names = declared_functions()
    assert len(names) >= 35
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing
    assert b"gfx950" in L.lsa_version()


def test_no_device_fails_loudly():
    """Without a HIP device context creation must fail, never fall back to a CPU path."""
    from laser_slam_amd import capi
    if capi.lib().lsa_device_count() > 0:
        pytest.skip("a GPU is present")
    p = capi.param_preset("LMS511")
    with pytest.raises(capi.LsaError):
        capi.Context(p)


def test_presets_and_validation():
    from laser_slam_amd import capi
    for name in ("LMS211", "LMS511", "LMS151"):
        p = capi.param_preset(name)
        assert p.points in (181, 361, 541) and p.min_range == 10
    with pytest.raises(capi.LsaError):
        capi.param_preset("HOKUYO_UTM_30LX")        # no such preset in the reference either


@pytest.mark.parametrize("name", list(SETS))
def test_tables_bit_exact(name):
    from laser_slam_amd import capi, synth
    g = load(name)
    laser = SETS[name][0]
    p = capi.param_preset(laser) if laser else capi.param_make(**synth.SYN1081)
    gp = golden_param(g)
    assert np.float32(p.dfi) == np.float32(gp["dfi"])
    p.fi_min = float(np.float32(gp["fi_min"]))
    if "tab_fi" in g:
        fi, si, co = capi.param_tables(p)
        assert same_bits(fi, g["tab_fi"]) and same_bits(si, g["tab_si"]) and same_bits(co, g["tab_co"])


@pytest.mark.parametrize("name", ["mit", "fr079", "intel", "indoor"])
def test_carmen_reader(name):
    from laser_slam_amd import capi
    g = load(name)
    laser, fn, tag = SETS[name]
    p = capi.param_preset(laser)
    r, x, y, bad, pose = capi.carmen_read(p, os.path.join(GOLDEN, fn), tag)
    assert same_bits(r, g["raw_r"]) and np.array_equal(bad, g["raw_bad"])
    gp = golden_param(g)
    assert p.max_range == int(gp["max_range"])
    assert np.float32(p.fi_min) == np.float32(gp["fi_min"]) and np.float32(p.fi_max) == np.float32(gp["fi_max"])
    if "reader_pose" in g:
        n = r.shape[1]
        assert same_bits(pose, g["reader_pose"])
        assert same_bits(x[:, :n - 1], g["reader_x"][:, :n - 1]) and same_bits(y[:, :n - 1], g["reader_y"][:, :n - 1])


def test_carmen_reader_errors(tmp_path):
    from laser_slam_amd import capi
    p = capi.param_preset("LMS511")
    with pytest.raises(capi.LsaError):
        capi.carmen_read(p, str(tmp_path / "missing.log"))
    bad = tmp_path / "wrong_beams.log"
    bad.write_text("ROBOTLASER1 0 -1.57 3.14 0.0087 50 0.01 0 5 1 2 3 4 5 0 0 0 0 0 0 0 0 0 0 0 0 0 h 0\n")
    with pytest.raises(capi.LsaError):
        capi.carmen_read(p, str(bad))               # beam count neither points nor points-1
    empty = tmp_path / "empty.log"
    empty.write_text("# nothing\nVERTEX2 0 0 0 0\n")
    with pytest.raises(capi.LsaError):
        capi.carmen_read(p, str(empty))
