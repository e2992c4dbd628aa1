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
