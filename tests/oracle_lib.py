"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, when present,
the reference built in place (oracle/_ref/*.so).

TEST INFRASTRUCTURE: imported only from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  Nothing under laser_slam_amd/ imports this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_ZHPSM_SO = os.path.join(ORACLE_DIR, "_ref", "libref_zhpsm.so")
REF_PMAP_SO = os.path.join(ORACLE_DIR, "_ref", "libref_pmap.so")

f32 = np.float32
i32 = np.int32


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def build_oracle():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle"])


def _load_oracle():
    if not os.path.exists(ORACLE_SO):
        build_oracle()
    L = C.CDLL(ORACLE_SO)
    L.orc_param_new.restype = C.c_void_p
    L.orc_param_new.argtypes = [C.c_int] * 7
    L.orc_param_preset.restype = C.c_void_p
    L.orc_param_preset.argtypes = [C.c_char_p]
    L.orc_param_free.argtypes = [C.c_void_p]
    L.orc_param_set_fimin_maxrange.argtypes = [C.c_void_p, C.c_float, C.c_int]
    L.orc_pmap_new.restype = C.c_void_p
    L.orc_pmap_new.argtypes = [C.c_int] * 4 + [C.c_float, C.c_float, C.c_double]
    return L


class _ParamStruct(C.Structure):
    _fields_ = [("points", C.c_int), ("fov_deg", C.c_int), ("max_range", C.c_int),
                ("min_range", C.c_int), ("min_valid", C.c_int), ("window", C.c_int),
                ("fi_min", C.c_float), ("fi_max", C.c_float), ("dfi", C.c_float),
                ("fi", C.POINTER(C.c_float)), ("si", C.POINTER(C.c_float)),
                ("co", C.POINTER(C.c_float))]


# The synthetic 1081-beam sensor has no preset in the reference;
# these values are injected identically into oracle, reference and GPU path.
SYN1081 = dict(points=1081, fov_deg=270, max_range=3000, min_valid=300, window=200,
               fi_min_deg=-135, fi_max_deg=135)


class Oracle:
    """CPU restatement; one instance per laser parameter block."""

    _L = None

    def __init__(self, laser=None, **custom):
        if Oracle._L is None:
            Oracle._L = _load_oracle()
        L = self.L = Oracle._L
        if laser is not None:
            self.h = L.orc_param_preset(laser.encode())
        else:
            self.h = L.orc_param_new(custom["points"], custom["fov_deg"], custom["max_range"],
                                     custom["min_valid"], custom["window"],
                                     custom["fi_min_deg"], custom["fi_max_deg"])
        if not self.h:
            raise ValueError("unknown laser preset %r" % laser)

    @property
    def param(self):
        return C.cast(self.h, C.POINTER(_ParamStruct)).contents

    @property
    def n(self):
        return self.param.points

    def param_dict(self):
        s = self.param
        return dict(points=s.points, fov_deg=s.fov_deg, max_range=s.max_range,
                    min_range=s.min_range, min_valid=s.min_valid, window=s.window,
                    fi_min=s.fi_min, fi_max=s.fi_max, dfi=s.dfi)

    def set_fimin_maxrange(self, fi_min, max_range):
        self.L.orc_param_set_fimin_maxrange(self.h, C.c_float(fi_min), int(max_range))

    def tables(self):
        n = self.n
        s = self.param
        return (np.ctypeslib.as_array(s.fi, (n,)).copy(), np.ctypeslib.as_array(s.si, (n,)).copy(),
                np.ctypeslib.as_array(s.co, (n,)).copy())

    def median(self, r):
        r = np.ascontiguousarray(r, f32).copy()
        self.L.orc_median(C.c_void_p(self.h), _p(r))
        return r

    def preprocess(self, r, bad=None):
        r = np.ascontiguousarray(r, f32).copy()
        bad = np.zeros(self.n, i32) if bad is None else np.ascontiguousarray(bad, i32).copy()
        seg = np.zeros(self.n, i32)
        self.L.orc_preprocess(C.c_void_p(self.h), _p(r), _p(bad), _p(seg))
        return r, bad, seg

    def scan_project(self, r, bad, seg, pose):
        new_r = np.zeros(self.n, f32)
        new_bad = np.zeros(self.n, i32)
        pose = np.ascontiguousarray(pose, f32)
        self.L.orc_scan_project(C.c_void_p(self.h), _p(r), _p(bad), _p(seg), _p(pose), _p(new_r), _p(new_bad))
        return new_r, new_bad

    def orientation_search(self, ref_r, ref_bad, new_r, new_bad):
        dth = C.c_float(0)
        st = self.L.orc_orientation_search(C.c_void_p(self.h), _p(ref_r), _p(ref_bad), _p(new_r), _p(new_bad),
                                           C.byref(dth))
        return st, dth.value

    def translation_estimation(self, ref_r, ref_bad, new_r, new_bad, Cw=4900.0):
        dx, dy, err = C.c_float(0), C.c_float(0), C.c_float(0)
        st = self.L.orc_translation_estimation(C.c_void_p(self.h), _p(ref_r), _p(ref_bad), _p(new_r), _p(new_bad),
                                               C.c_float(Cw), C.byref(dx), C.byref(dy), C.byref(err))
        return st, dx.value, dy.value, err.value

    def match(self, which, ref, act, ref_pose=(0, 0, 0), act_pose=(0, 0, 0)):
        """ref/act = (r, bad, seg) preprocessed.  Returns (status, pose[3], ret, iters)."""
        rp = np.asarray(ref_pose, f32)
        ap = np.asarray(act_pose, f32)
        out = np.zeros(3, f32)
        ret = C.c_float(0)
        it = C.c_int(0)
        st = self.L.orc_match(C.c_void_p(self.h), int(which), _p(ref[0]), _p(ref[1]), _p(rp),
                              _p(act[0]), _p(act[1]), _p(act[2]), _p(ap), _p(out), C.byref(ret), C.byref(it))
        return st, out, ret.value, it.value

    def error_index2(self, ref, act, ref_pose, act_pose):
        rp = np.asarray(ref_pose, f32)
        ap = np.asarray(act_pose, f32)
        exy = np.zeros(2, f32)
        na = C.c_int(0)
        self.L.orc_error_index2(C.c_void_p(self.h), _p(ref[0]), _p(ref[1]), _p(rp),
                                _p(act[0]), _p(act[1]), _p(act[2]), _p(ap), _p(exy), C.byref(na))
        return exy, na.value

    def fmatch(self, ref_raw, act_raw, first):
        """raw = (r, bad, seg) reader-format; returns (pose, pair, ref', act')."""
        ref = [np.ascontiguousarray(a).copy() for a in ref_raw]
        act = [np.ascontiguousarray(a).copy() for a in act_raw]
        pose = np.zeros(3, f32)
        pair = np.zeros(2, f32)
        self.L.orc_fmatch(C.c_void_p(self.h), _p(ref[0]), _p(ref[1]), _p(ref[2]),
                          _p(act[0]), _p(act[1]), _p(act[2]), int(bool(first)), _p(pose), _p(pair))
        return pose, pair, ref, act

    def match_batch(self, which, r, bad, seg, ref_idx, act_idx):
        B = len(ref_idx)
        pose = np.zeros((B, 3), f32)
        st = np.zeros(B, i32)
        ri = np.ascontiguousarray(ref_idx, i32)
        ai = np.ascontiguousarray(act_idx, i32)
        self.L.orc_match_batch(C.c_void_p(self.h), int(which), B, _p(r), _p(bad), _p(seg), _p(ri), _p(ai),
                               _p(pose), _p(st))
        return pose, st

    def read_carmen(self, filename, tag="ROBOTLASER1"):
        ptrs = [C.c_void_p() for _ in range(5)]
        S = self.L.orc_read_carmen(C.c_void_p(self.h), filename.encode(), tag.encode(),
                                   *[C.byref(p) for p in ptrs])
        if S == 0:
            raise IOError("orc_read_carmen failed on %s" % filename)
        n = self.n
        outs = []
        for p, (dt, w) in zip(ptrs, [(f32, n), (f32, n), (f32, n), (i32, n), (f32, 3)]):
            ct = C.c_float if dt is f32 else C.c_int
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (S, w)).copy()
            self.L.orc_free(p)
            outs.append(a)
        return outs  # r, x, y, bad, pose

    def oplus(self, a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); o = np.zeros(3)
        self.L.orc_oplus(_p(a), _p(b), _p(o))
        return o

    def ominus(self, a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); o = np.zeros(3)
        self.L.orc_ominus(_p(a), _p(b), _p(o))
        return o


class _PmapStruct(C.Structure):
    _fields_ = [("size_x", C.c_int), ("size_y", C.c_int), ("center_x", C.c_int), ("center_y", C.c_int),
                ("off_x", C.c_float), ("off_y", C.c_float), ("res", C.c_double),
                ("hit", C.POINTER(C.c_float)), ("sum", C.POINTER(C.c_float)), ("prob", C.POINTER(C.c_float))]


class OraclePMap:
    def __init__(self, sx, sy, cx, cy, offx, offy, res):
        if Oracle._L is None:
            Oracle._L = _load_oracle()
        self.L = Oracle._L
        self.h = self.L.orc_pmap_new(sx, sy, cx, cy, C.c_float(offx), C.c_float(offy), C.c_double(res))
        self.shape = (sx, sy)

    def cell(self, px, py):
        ix, iy = C.c_int(0), C.c_int(0)
        self.L.orc_pmap_cell(C.c_void_p(self.h), C.c_float(px), C.c_float(py), C.byref(ix), C.byref(iy))
        return ix.value, iy.value

    def update(self, gx, gy, rx, ry):
        gx = np.ascontiguousarray(gx, f32); gy = np.ascontiguousarray(gy, f32)
        self.L.orc_pmap_update(C.c_void_p(self.h), _p(gx), _p(gy), len(gx), C.c_float(rx), C.c_float(ry))

    def compute(self):
        self.L.orc_pmap_compute(C.c_void_p(self.h))

    def layers(self):
        s = C.cast(self.h, C.POINTER(_PmapStruct)).contents
        n = self.shape[0] * self.shape[1]
        return tuple(np.ctypeslib.as_array(p, (n,)).reshape(self.shape).copy() for p in (s.hit, s.sum, s.prob))

    def cloud(self, thre, cap=1 << 22):
        x = np.zeros(cap, f32); y = np.zeros(cap, f32)
        n = self.L.orc_pmap_cloud(C.c_void_p(self.h), C.c_float(thre), _p(x), _p(y), cap)
        return x[:n], y[:n]

    def __del__(self):
        try:
            self.L.orc_pmap_free(C.c_void_p(self.h))
        except Exception:
            pass


def grid_line(sx, sy, ex, ey, lib=None, fn="orc_grid_line"):
    if lib is None:
        if Oracle._L is None:
            Oracle._L = _load_oracle()
        lib = Oracle._L
    cap = max(abs(ex - sx), abs(ey - sy)) + 8
    xs = np.zeros(cap, i32); ys = np.zeros(cap, i32)
    n = getattr(lib, fn)(sx, sy, ex, ey, _p(xs), _p(ys), cap)
    return np.stack([xs[:n], ys[:n]], 1)


def likelihood_field(prob, res, max_corr=1.0, free_thr=0.3, unknown=0.5):
    if Oracle._L is None:
        Oracle._L = _load_oracle()
    prob = np.ascontiguousarray(prob, f32)
    out = np.zeros_like(prob)
    Oracle._L.orc_likelihood_field(_p(prob), prob.shape[0], prob.shape[1], C.c_double(res), C.c_float(max_corr),
                                   C.c_float(free_thr), C.c_float(unknown), _p(out))
    return out


def score_poses(lik, cx, cy, offx, offy, res, scan_x, scan_y, decim, poses, max_corr=1.0, std_frame=False):
    if Oracle._L is None:
        Oracle._L = _load_oracle()
    lik = np.ascontiguousarray(lik, f32)
    sx_, sy_ = np.ascontiguousarray(scan_x, f32), np.ascontiguousarray(scan_y, f32)
    poses = np.ascontiguousarray(poses, f32)
    out = np.zeros(len(poses), f32)
    fn = Oracle._L.orc_score_poses_std if std_frame else Oracle._L.orc_score_poses
    fn(_p(lik), lik.shape[0], lik.shape[1], cx, cy, C.c_float(offx), C.c_float(offy),
       C.c_double(res), C.c_float(max_corr), _p(sx_), _p(sy_), len(sx_), decim, _p(poses),
       len(poses), _p(out))
    return out


def systematic_resample(w, u0, n_out):
    if Oracle._L is None:
        Oracle._L = _load_oracle()
    w = np.ascontiguousarray(w, f32)
    parent = np.zeros(n_out, i32)
    Oracle._L.orc_systematic_resample(_p(w), len(w), C.c_double(u0), n_out, _p(parent))
    return parent


def valid_pose(prob, geom, px, py, free_thr=0.3, unknown=0.5):
    """CVPmap::isValidPose; geom = (size_x, size_y, center_x, center_y, off_x, off_y, res)."""
    prob = np.ascontiguousarray(prob, f32)
    sx, sy, cx, cy, ox, oy, res = geom
    return _orc().orc_valid_pose(_p(prob), int(sx), int(sy), int(cx), int(cy), C.c_float(ox), C.c_float(oy), C.c_double(res),
                                 C.c_float(px), C.c_float(py), C.c_float(free_thr), C.c_float(unknown))


def standard_resample(w, u):
    """Standard_resampler + copy_resamples: parents in ascending order, and the uniforms left over (0 if fine)."""
    if Oracle._L is None:
        Oracle._L = _load_oracle()
    w = np.ascontiguousarray(w, f32); u = np.ascontiguousarray(u, np.float64)
    parent = np.zeros(len(u), i32)
    Oracle._L.orc_standard_resample.restype = C.c_int
    left = Oracle._L.orc_standard_resample(_p(w), len(w), _p(u), len(u), _p(parent))
    return parent, left


def p2p_icp(rx, ry, cx, cy, init, max_iter=100, thr_d=0.75, thr_a=np.deg2rad(10.0), alfa=0.5, smallest=0.05):
    if Oracle._L is None:
        Oracle._L = _load_oracle()
    rx, ry, cx, cy = (np.ascontiguousarray(a, f32) for a in (rx, ry, cx, cy))
    init = np.ascontiguousarray(init, np.float64)
    out = np.zeros(16, np.float64)
    Oracle._L.orc_p2p_icp(_p(rx), _p(ry), len(rx), _p(cx), _p(cy), len(cx), _p(init), int(max_iter),
                          C.c_float(thr_d), C.c_float(np.float32(thr_a)), C.c_float(alfa), C.c_float(smallest), _p(out))
    return out


def neff(w):
    if Oracle._L is None:
        Oracle._L = _load_oracle()
    w = np.ascontiguousarray(w, f32)
    Oracle._L.orc_neff.restype = C.c_double
    return Oracle._L.orc_neff(_p(w), len(w))


# ---------------------------------------------------------------------------
# reference built in place (only in the build container / when _ref travelled)

def have_ref():
    return os.path.exists(REF_ZHPSM_SO) and os.path.exists(REF_PMAP_SO)


class Ref:
    _L = None

    def __init__(self, laser=None, **custom):
        if Ref._L is None:
            L = C.CDLL(REF_ZHPSM_SO)
            L.ref_pm_create.restype = C.c_void_p
            L.ref_pm_create.argtypes = [C.c_char_p]
            L.ref_pm_create_custom.restype = C.c_void_p
            L.ref_pm_create_custom.argtypes = [C.c_char_p] + [C.c_int] * 7
            L.ref_pm_set_fimin_maxrange.argtypes = [C.c_void_p, C.c_float, C.c_int]
            Ref._L = L
        L = self.L = Ref._L
        if laser is not None:
            self.h = L.ref_pm_create(laser.encode())
        else:
            self.h = L.ref_pm_create_custom(b"SYN", custom["points"], custom["fov_deg"], custom["max_range"],
                                            custom["min_valid"], custom["window"],
                                            custom["fi_min_deg"], custom["fi_max_deg"])
        if not self.h:
            raise ValueError("reference rejected laser %r" % laser)
        self.n = self.param_dict()["points"]

    def param_dict(self):
        ip = (C.c_int * 6)(); fp = (C.c_float * 3)()
        self.L.ref_pm_get_param(C.c_void_p(self.h), ip, fp)
        return dict(points=ip[0], fov_deg=ip[1], max_range=ip[2], min_range=ip[3], min_valid=ip[4],
                    window=ip[5], fi_min=fp[0], fi_max=fp[1], dfi=fp[2])

    def set_fimin_maxrange(self, fi_min, max_range):
        self.L.ref_pm_set_fimin_maxrange(C.c_void_p(self.h), C.c_float(fi_min), int(max_range))

    def tables(self):
        fi, si, co = (np.zeros(self.n, f32) for _ in range(3))
        self.L.ref_pm_get_tables(C.c_void_p(self.h), _p(fi), _p(si), _p(co))
        return fi, si, co

    def read_carmen(self, filename, tag="ROBOTLASER1"):
        S = self.L.ref_read_carmen(C.c_void_p(self.h), filename.encode(), tag.encode())
        if S == 0:
            raise IOError("reference readCarmon failed")
        n = self.n
        r = np.zeros((S, n), f32); x = np.zeros((S, n), f32); y = np.zeros((S, n), f32)
        bad = np.zeros((S, n), i32); pose = np.zeros((S, 3), f32)
        for k in range(S):
            self.L.ref_get_scan(C.c_void_p(self.h), k, _p(r[k]), _p(x[k]), _p(y[k]), _p(bad[k]), _p(pose[k]))
        return r, x, y, bad, pose

    def median(self, r):
        r = np.ascontiguousarray(r, f32).copy()
        self.L.ref_median(C.c_void_p(self.h), _p(r))
        return r

    def preprocess(self, r, bad=None):
        r = np.ascontiguousarray(r, f32).copy()
        bad = np.zeros(self.n, i32) if bad is None else np.ascontiguousarray(bad, i32).copy()
        seg = np.zeros(self.n, i32)
        self.L.ref_preprocess(C.c_void_p(self.h), _p(r), _p(bad), _p(seg))
        return r, bad, seg

    def scan_project(self, r, bad, seg, pose):
        new_r = np.zeros(self.n, f32); new_bad = np.zeros(self.n, i32)
        pose = np.ascontiguousarray(pose, f32)
        self.L.ref_scan_project(C.c_void_p(self.h), _p(r), _p(bad), _p(seg), _p(pose), _p(new_r), _p(new_bad))
        return new_r, new_bad

    def orientation_search(self, ref_r, ref_bad, new_r, new_bad):
        dth = C.c_float(0)
        st = self.L.ref_orientation_search(C.c_void_p(self.h), _p(ref_r), _p(ref_bad), _p(new_r), _p(new_bad),
                                           C.byref(dth))
        return st, dth.value

    def translation_estimation(self, ref_r, ref_bad, new_r, new_bad, Cw=4900.0):
        dx, dy, err = C.c_float(0), C.c_float(0), C.c_float(0)
        st = self.L.ref_translation_estimation(C.c_void_p(self.h), _p(ref_r), _p(ref_bad), _p(new_r), _p(new_bad),
                                               C.c_float(Cw), C.byref(dx), C.byref(dy), C.byref(err))
        return st, dx.value, dy.value, err.value

    def match(self, which, ref, act, ref_pose=(0, 0, 0), act_pose=(0, 0, 0)):
        rp = np.asarray(ref_pose, f32); ap = np.asarray(act_pose, f32)
        out = np.zeros(3, f32); ret = np.zeros(1, f32)
        st = self.L.ref_match(C.c_void_p(self.h), int(which), _p(ref[0]), _p(ref[1]), _p(ref[2]), _p(rp),
                              _p(act[0]), _p(act[1]), _p(act[2]), _p(ap), _p(out), _p(ret))
        return st, out, float(ret[0])

    def error_index2(self, ref, act, ref_pose, act_pose):
        rp = np.asarray(ref_pose, f32); ap = np.asarray(act_pose, f32)
        exy = np.zeros(2, f32); na = C.c_int(0)
        self.L.ref_error_index2(C.c_void_p(self.h), _p(ref[0]), _p(ref[1]), _p(ref[2]), _p(rp),
                                _p(act[0]), _p(act[1]), _p(act[2]), _p(ap), _p(exy), C.byref(na))
        return exy, na.value

    def fmatch(self, ref_raw, act_raw, first):
        ref = [np.ascontiguousarray(a).copy() for a in ref_raw]
        act = [np.ascontiguousarray(a).copy() for a in act_raw]
        pose = np.zeros(3, f32); pair = np.zeros(2, f32)
        self.L.ref_fmatch(C.c_void_p(self.h), _p(ref[0]), _p(ref[1]), _p(ref[2]),
                          _p(act[0]), _p(act[1]), _p(act[2]), int(bool(first)), _p(pose), _p(pair))
        return pose, pair, ref, act

    def oplus(self, a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); o = np.zeros(3)
        self.L.ref_oplus(_p(a), _p(b), _p(o))
        return o

    def ominus(self, a, b):
        a = np.asarray(a, np.float64); b = np.asarray(b, np.float64); o = np.zeros(3)
        self.L.ref_ominus(_p(a), _p(b), _p(o))
        return o


class RefPMap:
    _L = None

    def __init__(self, sx, sy, cx, cy, offx, offy, res, thre=0.95):
        if RefPMap._L is None:
            L = C.CDLL(REF_PMAP_SO)
            L.ref_pmap_create.restype = C.c_void_p
            L.ref_pmap_create.argtypes = [C.c_float]
            L.ref_pmap_init.argtypes = [C.c_void_p] + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_double]
            L.ref_pmap_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_float] * 3
            L.ref_pmap_cell.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
            RefPMap._L = L
        self.L = RefPMap._L
        self.h = self.L.ref_pmap_create(thre)
        self.L.ref_pmap_init(self.h, sx, sy, cx, cy, offx, offy, res)
        self.shape = (sx, sy)

    def cell(self, px, py):
        ix, iy = C.c_int(0), C.c_int(0)
        self.L.ref_pmap_cell(self.h, px, py, C.byref(ix), C.byref(iy))
        return ix.value, iy.value

    def update(self, gx, gy, rx, ry):
        gx = np.ascontiguousarray(gx, f32); gy = np.ascontiguousarray(gy, f32)
        self.L.ref_pmap_update(self.h, _p(gx), _p(gy), len(gx), rx, ry, 0.0)

    def compute(self):
        self.L.ref_pmap_compute(C.c_void_p(self.h))

    def layers(self):
        out = []
        for layer in range(3):
            a = np.zeros(self.shape, f32)
            self.L.ref_pmap_get(C.c_void_p(self.h), layer, _p(a))
            out.append(a)
        return tuple(out)

    def cloud(self, cap=1 << 22):
        x = np.zeros(cap, f32); y = np.zeros(cap, f32)
        n = self.L.ref_pmap_cloud(C.c_void_p(self.h), _p(x), _p(y), cap)
        return x[:n], y[:n]

    def __del__(self):
        try:
            self.L.ref_pmap_destroy(C.c_void_p(self.h))
        except Exception:
            pass


def _orc():
    if Oracle._L is None:
        Oracle._L = _load_oracle()
    return Oracle._L


def simulate_scan(prob, cx, cy, offx, offy, res, pose, n_beams, fov, min_angle, max_range):
    """CVPmap::laserScanSimulator for one pose -> (ranges [m], bad)."""
    prob = np.ascontiguousarray(prob, f32)
    r = np.zeros(n_beams, f32); bad = np.zeros(n_beams, i32)
    pose = np.ascontiguousarray(pose, f32)
    _orc().orc_simulate_scan(_p(prob), prob.shape[0], prob.shape[1], cx, cy, C.c_float(offx), C.c_float(offy),
                                C.c_double(res), _p(pose), n_beams, C.c_float(fov), C.c_float(min_angle),
                                C.c_float(max_range), _p(r), _p(bad))
    return r, bad


def beam_model(prob, cx, cy, offx, offy, res, poses, obs_r, obs_bad, fov, min_angle, max_range):
    prob = np.ascontiguousarray(prob, f32); poses = np.ascontiguousarray(poses, f32)
    obs_r = np.ascontiguousarray(obs_r, f32); obs_bad = np.ascontiguousarray(obs_bad, i32)
    score = np.zeros(len(poses), f32)
    _orc().orc_beam_model(_p(prob), prob.shape[0], prob.shape[1], cx, cy, C.c_float(offx), C.c_float(offy),
                             C.c_double(res), _p(poses), len(poses), _p(obs_r), _p(obs_bad), len(obs_r),
                             C.c_float(fov), C.c_float(min_angle), C.c_float(max_range), _p(score))
    return score


def topk(score, k):
    score = np.ascontiguousarray(score, f32)
    idx = np.zeros(k, i32)
    _orc().orc_topk(_p(score), len(score), k, _p(idx))
    return idx


def pf_estimate(poses, score, index):
    poses = np.ascontiguousarray(poses, f32); score = np.ascontiguousarray(score, f32)
    index = np.ascontiguousarray(index, i32)
    mean = np.zeros(3, np.float64); mw, md = C.c_double(0), C.c_double(0)
    _orc().orc_pf_estimate(_p(poses), _p(score), _p(index), len(index), _p(mean), C.byref(mw), C.byref(md))
    return mean, mw.value, md.value


def pf_predict(n, mean, sigma, normals):
    mean = np.ascontiguousarray(mean, np.float64); sigma = np.ascontiguousarray(sigma, np.float64)
    normals = np.ascontiguousarray(normals, f32); poses = np.zeros((n, 3), f32)
    _orc().orc_pf_predict(n, _p(mean), _p(sigma), _p(normals), _p(poses))
    return poses


def pf_motion(poses, delta, sigma, normals):
    poses = np.ascontiguousarray(poses, f32).copy(); normals = np.ascontiguousarray(normals, f32)
    delta = np.ascontiguousarray(delta, np.float64); sigma = np.ascontiguousarray(sigma, np.float64)
    _orc().orc_pf_motion(len(poses), _p(delta), _p(sigma), _p(normals), _p(poses))
    return poses


def pf_roughen(poses, K, normals):
    poses = np.ascontiguousarray(poses, f32).copy(); normals = np.ascontiguousarray(normals, f32)
    _orc().orc_pf_roughen(len(poses), C.c_double(K), _p(normals), _p(poses))
    return poses


def icp_goodness(prob, cx, cy, offx, offy, res, poses, obs_x, obs_y, fov, min_angle, max_range):
    prob = np.ascontiguousarray(prob, f32); poses = np.ascontiguousarray(poses, f32)
    obs_x = np.ascontiguousarray(obs_x, f32); obs_y = np.ascontiguousarray(obs_y, f32)
    w = np.zeros(len(poses), f32); rel = np.zeros((len(poses), 3), f32)
    _orc().orc_icp_goodness(_p(prob), prob.shape[0], prob.shape[1], cx, cy, C.c_float(offx), C.c_float(offy),
                            C.c_double(res), _p(poses), len(poses), _p(obs_x), _p(obs_y), len(obs_x),
                            C.c_float(fov), C.c_float(min_angle), C.c_float(max_range), _p(w), _p(rel))
    return w, rel


class _PlicpParams(C.Structure):
    _fields_ = [("max_angular_correction_deg", C.c_double), ("max_linear_correction", C.c_double), ("max_iterations", C.c_int),
                ("epsilon_xy", C.c_double), ("epsilon_theta", C.c_double), ("max_correspondence_dist", C.c_double),
                ("restart", C.c_int), ("restart_threshold_mean_error", C.c_double), ("restart_dt", C.c_double),
                ("restart_dtheta", C.c_double), ("outliers_maxPerc", C.c_double), ("outliers_adaptive_order", C.c_double),
                ("outliers_adaptive_mult", C.c_double), ("outliers_remove_doubles", C.c_int)]


def plicp(ref_r_cm, cur_r_cm, fi_min, dfi, min_range_cm, max_range_cm):
    """CCanonicalMatcher's point-to-line ICP (oracle/plicp_oracle.c): 8-double result record."""
    P = _PlicpParams()
    _orc().orc_plicp_params_default(C.byref(P))
    a = np.ascontiguousarray(ref_r_cm, f32); b = np.ascontiguousarray(cur_r_cm, f32)
    out = np.zeros(8, np.float64)
    _orc().orc_plicp(_p(a), _p(b), len(a), C.c_float(fi_min), C.c_float(dfi), int(min_range_cm), int(max_range_cm),
                     C.byref(P), _p(out))
    return out


def plicp_cov(ref_r_cm, cur_r_cm, fi_min, dfi, min_range_cm, max_range_cm, sigma=0.04):
    """orc_plicp_cov: result record, sm_icp-style covariance (3x3) and the correspondences it was formed over."""
    P = _PlicpParams()
    _orc().orc_plicp_params_default(C.byref(P))
    a = np.ascontiguousarray(ref_r_cm, f32); b = np.ascontiguousarray(cur_r_cm, f32)
    out = np.zeros(8, np.float64); cov = np.zeros(9, np.float64); corr = np.zeros(2 * len(a), np.int32)
    _orc().orc_plicp_cov(_p(a), _p(b), len(a), C.c_float(fi_min), C.c_float(dfi), int(min_range_cm), int(max_range_cm),
                         C.byref(P), C.c_double(sigma), _p(out), _p(cov), _p(corr))
    return out, cov.reshape(3, 3), corr.reshape(-1, 2)
