"""CPU: laser_slam_amd/csrc/lsa_math.h (the FP32 sin/cos/atan2 the kernels use) against
the container's libm, bit for bit.  Strided here to stay fast; `math_bits_check 1` is
the exhaustive run (every float with |x| < 120; every float for atanf) recorded in DESIGN.md."""
import json
import os
import subprocess

from helpers import ROOT


def test_device_math_matches_libm(tmp_path):
    exe = str(tmp_path / "math_bits_check")
    src = os.path.join(ROOT, "tests", "native", "math_bits_check.cpp")
    subprocess.check_call(["g++", "-O2", "-mfma", "-ffp-contract=off", "-pthread", "-o", exe, src])
    out = subprocess.run([exe, "61"], capture_output=True, text=True)
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert out.returncode == 0, out.stderr[-2000:]
    assert res["sin"] == 0 and res["cos"] == 0 and res["atan"] == 0 and res["atan2"] == 0
