"""Shared test helpers (golden loading, parameter blocks)."""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# golden set name -> (laser preset or None for the injected 1081-beam block, log head, tag)
SETS = {
    "mit": ("LMS511", "mit_head.log", "ROBOTLASER1"),
    "fr079": ("LMS511", "fr079_head.log", "ROBOTLASER1"),
    "intel": ("LMS211", "intel_head.log", "ROBOTLASER1"),
    "indoor": ("LMS151", "indoor_head.txt", "LMS151"),
    "syn1081": (None, None, None),
}


def load(name):
    z = np.load(os.path.join(GOLDEN, "zhpsm_%s.npz" % name))
    return {k: z[k] for k in z.files}


def golden_param(g):
    return dict(zip([str(k) for k in g["param_keys"]], g["param"]))


def same_bits(a, b):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    return np.array_equal(a.view(np.uint8), b.view(np.uint8)) or np.array_equal(a, b, equal_nan=True)


def chain_poses(rel_cm_rad, failed):
    """Global laser poses (metres / rad, double) of a scan-to-scan chain: pose_k = pose_{k-1} (+) rel_k, where a failed
    match contributes no motion (OrientedPoint2D::oplus, src/zhpsm/point.cpp:59-65).  Shared by the generator of
    tests/golden/full_*.npz and the GPU test so that both build the same map inputs."""
    S = len(rel_cm_rad) + 1
    out = np.zeros((S, 3), np.float64)
    for k in range(1, S):
        x, y, th = out[k - 1]
        rx, ry, rt = (0.0, 0.0, 0.0) if failed[k - 1] else (float(rel_cm_rad[k - 1][0]) / 100.0, float(rel_cm_rad[k - 1][1]) / 100.0,
                                                             float(rel_cm_rad[k - 1][2]))
        c, s = np.cos(th), np.sin(th)
        out[k] = (x + rx * c - ry * s, y + rx * s + ry * c, th + rt)
    return out


def scan_endpoints(r_cm, bad, fi, pose, max_range_cm):
    """Endpoints (global metres, float32) of the usable readings of one preprocessed scan at `pose`, padded to the
    scan's width, and their count - the arrays CPMap::updateMap takes (src/drawmap/PMap.cpp:46)."""
    ok = (bad == 0) & (r_cm <= max_range_cm)
    rr = r_cm[ok].astype(np.float64) / 100.0
    gx = (pose[0] + rr * np.cos(pose[2] + fi[ok].astype(np.float64))).astype(np.float32)
    gy = (pose[1] + rr * np.sin(pose[2] + fi[ok].astype(np.float64))).astype(np.float32)
    px, py = np.zeros(len(r_cm), np.float32), np.zeros(len(r_cm), np.float32)
    px[:len(gx)] = gx; py[:len(gy)] = gy
    return px, py, len(gx)
