"""CPU: the reference's own drivers compile, link and run against the drop-in headers and library.

Runs only where /root/reference exists (the build container).  The reference sources are copied to a temporary
directory at test time (never into the repository) so that their quote-includes resolve to
laser_slam_amd/host/<module>, which is how a maintainer switches the build over."""
import os
import shutil
import subprocess

import pytest

from helpers import ROOT

REF = "/root/reference/src"
HOST = os.path.join(ROOT, "laser_slam_amd", "host")
LIBDIR = os.path.join(ROOT, "laser_slam_amd", "lib")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "zhpsm")), reason="reference tree not present")


def _cc(args, cwd):
    res = subprocess.run(["g++", "-std=c++11", "-w"] + args, cwd=cwd, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-4000:]


def _stage(tmp_path, module, names):
    for nm in names:
        shutil.copy(os.path.join(REF, module, nm), tmp_path / nm)


def test_zhpsm_test_cpp_compiles_and_links(tmp_path):
    """src/zhpsm/test.cpp: CPolarMatch("LMS511").run511Data(file, -1) - the driver VERDICT r1 found not to compile."""
    _stage(tmp_path, "zhpsm", ["test.cpp"])
    exe = str(tmp_path / "zhpsm_test")
    _cc(["-I", os.path.join(HOST, "zhpsm"), "-I", os.path.join(ROOT, "include"), "test.cpp", "-o", exe,
         "-L", LIBDIR, "-llsa_host", "-llsa_core", "-Wl,-rpath," + LIBDIR], tmp_path)
    assert os.path.exists(exe)


def test_zhpsm_testpose_same_output_as_reference(tmp_path):
    """src/zhpsm/testPose.cpp (operator+, oplus / ominus, operator<<) prints the same against the drop-in's point.{h,cpp}
    as against the reference's own (compiled where they lie)."""
    _stage(tmp_path, "zhpsm", ["testPose.cpp"])
    ours, theirs = str(tmp_path / "pose_ours"), str(tmp_path / "pose_ref")
    _cc(["-I", os.path.join(HOST, "zhpsm"), "testPose.cpp", os.path.join(HOST, "zhpsm", "point.cpp"), "-o", ours], tmp_path)
    ref_dir = tmp_path / "refbuild"
    ref_dir.mkdir()
    shutil.copy(os.path.join(REF, "zhpsm", "testPose.cpp"), ref_dir / "testPose.cpp")
    _cc(["-I", os.path.join(REF, "zhpsm"), "testPose.cpp", os.path.join(REF, "zhpsm", "point.cpp"), "-o", theirs], ref_dir)
    a = subprocess.run([ours], capture_output=True, text=True, timeout=60)
    b = subprocess.run([theirs], capture_output=True, text=True, timeout=60)
    assert a.returncode == 0 and b.returncode == 0
    assert a.stdout == b.stdout and "p22:" in a.stdout and "p23:" in a.stdout


def test_zhcsm_test_cpp_compiles_and_links(tmp_path):
    """src/zhcsm/test.cpp: CCanonicalMatcher().compareCSM2PSM_RawSeed(file, 1000) (the reference's header needs CSM and
    Eigen; the drop-in's needs neither)."""
    _stage(tmp_path, "zhcsm", ["test.cpp"])
    exe = str(tmp_path / "zhcsm_test")
    _cc(["-I", os.path.join(HOST, "zhcsm"), "-I", os.path.join(HOST, "zhpsm"), "-I", os.path.join(ROOT, "include"),
         "test.cpp", "-o", exe, "-L", LIBDIR, "-llsa_host", "-llsa_core", "-Wl,-rpath," + LIBDIR], tmp_path)
    assert os.path.exists(exe)


def test_point_helpers_used_by_tcp_slam(tmp_path):
    """deg2rad / rad2deg (src/tcp_slam/clientFrontend.cpp:226,318,366) and the operators of point.h:91-125, against the
    reference's header on the same expressions."""
    src = r'''
#include "point.h"
#include <cstdio>
using namespace zhpsm;
int main() {
    OrientedPoint2D a(1.5, -2.0, 3.0), b(0.25, 4.0, 2.5);
    OrientedPoint2D s = a + b, d = a - b;
    Point2D p = Point2D(1, 2) * 3.0, q = 2.0 * Point2D(1, 2);
    printf("%a %a %a %a %a %a %a %a %a %a %a %a %a\n", deg2rad(33.0), rad2deg(0.7), s.x, s.y, s.theta, d.x, d.y, d.theta,
           p.x, q.y, Point2D(1, 2) * Point2D(3, 4), (Point2D(1, 2) - Point2D(3, 5)).x, (Point2D(1, 2) + Point2D(3, 5)).y);
    std::cout << a << " " << Point2D(1, 2) << std::endl;
    return 0;
}
'''
    (tmp_path / "helpers.cpp").write_text(src)
    ours, theirs = str(tmp_path / "h_ours"), str(tmp_path / "h_ref")
    _cc(["-I", os.path.join(HOST, "zhpsm"), "helpers.cpp", os.path.join(HOST, "zhpsm", "point.cpp"), "-o", ours], tmp_path)
    _cc(["-I", os.path.join(REF, "zhpsm"), "helpers.cpp", os.path.join(REF, "zhpsm", "point.cpp"), "-o", theirs], tmp_path)
    a = subprocess.run([ours], capture_output=True, text=True, timeout=60)
    b = subprocess.run([theirs], capture_output=True, text=True, timeout=60)
    assert a.returncode == 0 and a.stdout == b.stdout and a.stdout.count("\n") == 2


def test_pf_localization_main_compiles_and_links(tmp_path):
    """src/pf_localization_main.cpp (MRPT's pf-localization application) compiles unchanged against
    the MRPT-shaped facade laser_slam_amd/host/mrptpf and links with the product libraries; tests/test_gpu_mrptpf.py
    runs the binary the oracle Makefile builds the same way."""
    if not os.path.exists(os.path.join(REF, "pf_localization_main.cpp")):
        pytest.skip("pf_localization_main.cpp not in the reference tree")
    shutil.copy(os.path.join(REF, "pf_localization_main.cpp"), tmp_path / "pf_localization_main.cpp")
    exe = str(tmp_path / "pf_localization")
    _cc(["-I", os.path.join(HOST, "mrptpf"), "-I", os.path.join(ROOT, "include"), "pf_localization_main.cpp", "-o", exe,
         "-L", LIBDIR, "-llsa_host", "-llsa_core", "-Wl,-rpath," + LIBDIR], tmp_path)
    assert os.path.exists(exe)
    # the copy in src/mrptpf differs from it in one comment only and compiles the same way
    twin = os.path.join(REF, "mrptpf", "pf_localization_main.cpp")
    if os.path.exists(twin):
        d = tmp_path / "twin"
        d.mkdir()
        shutil.copy(twin, d / "pf_localization_main.cpp")
        _cc(["-fsyntax-only", "-I", os.path.join(HOST, "mrptpf"), "-I", os.path.join(ROOT, "include"), "pf_localization_main.cpp"], d)
