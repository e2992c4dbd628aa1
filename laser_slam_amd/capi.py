"""ctypes binding of the C ABI (include/lsa_core.h) for tests and bench.py.

This is plumbing: every compute call goes straight into liblsa_core.so (HIP).
There is no Python or CPU implementation behind it — if the library is missing
or no GPU is present, calls raise.
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "liblsa_core.so")

MATCH_PSM, MATCH_ICP = 0, 1
ST_OK, ST_FEW_POINTS, ST_SINGULAR, ST_NO_ORIENT = 0, 1, 2, 3

vp = C.c_void_p


class LaserParam(C.Structure):
    _fields_ = [("points", C.c_int32), ("fov_deg", C.c_int32), ("max_range", C.c_int32),
                ("min_range", C.c_int32), ("min_valid", C.c_int32), ("window", C.c_int32),
                ("fi_min", C.c_float), ("fi_max", C.c_float), ("dfi", C.c_float)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class PfInfo(C.Structure):
    """lsa_pf_info (include/lsa_core.h)"""
    _fields_ = [("best_score", C.c_float), ("has_best_pose", C.c_int32), ("best_index", C.c_int64), ("best_pose", C.c_float * 3),
                ("resampled", C.c_int32), ("weight_sum", C.c_double), ("weight_sq_sum", C.c_double), ("n_eff", C.c_double)]

    def as_dict(self):
        return dict(best_score=self.best_score, best_index=self.best_index, best_pose=tuple(self.best_pose) if self.has_best_pose else None,
                    weight_sum=self.weight_sum, weight_sq_sum=self.weight_sq_sum, n_eff=self.n_eff, resampled=bool(self.resampled))


PF_RECORD_BYTES = 48


class LsaError(RuntimeError):
    pass


_lib = None


def lib():
    """Load liblsa_core.so (raises if it has not been built: no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch carries its own HIP runtime; when both live in one process it has to be loaded first, or this library's
    # runtime comes up without a device.  (The binding is test and bench plumbing: those processes use torch tensors.)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    if not os.path.exists(LIB_PATH):
        raise LsaError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'`" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.lsa_version.restype = C.c_char_p
    L.lsa_last_error.restype = C.c_char_p
    L.lsa_ctx_get_stream.restype = vp
    L.lsa_ctx_get_stream.argtypes = [vp]
    L.lsa_ctx_destroy.argtypes = [vp]
    L.lsa_ctx_destroy.restype = None
    L.lsa_grid_destroy.argtypes = [vp]
    L.lsa_grid_destroy.restype = None
    L.lsa_host_free.argtypes = [vp]
    L.lsa_host_free.restype = None
    L.lsa_param_make.argtypes = [C.POINTER(LaserParam)] + [C.c_int] * 7
    L.lsa_param_preset.argtypes = [C.POINTER(LaserParam), C.c_char_p]
    L.lsa_param_tables.argtypes = [C.POINTER(LaserParam), vp, vp, vp]
    L.lsa_ctx_create.argtypes = [C.POINTER(vp), C.c_int, C.POINTER(LaserParam)]
    L.lsa_ctx_set_param.argtypes = [vp, C.POINTER(LaserParam)]
    L.lsa_ctx_get_param.argtypes = [vp, C.POINTER(LaserParam)]
    L.lsa_ctx_set_stream.argtypes = [vp, vp]
    L.lsa_ctx_sync.argtypes = [vp]
    L.lsa_malloc.argtypes = [C.POINTER(vp), C.c_uint64]
    L.lsa_free.argtypes = [vp]
    L.lsa_memcpy_h2d.argtypes = [vp, vp, vp, C.c_uint64]
    L.lsa_memcpy_d2h.argtypes = [vp, vp, vp, C.c_uint64]
    L.lsa_memset.argtypes = [vp, vp, C.c_int, C.c_uint64]
    L.lsa_carmen_read.argtypes = [C.POINTER(LaserParam), C.c_char_p, C.c_char_p, C.POINTER(C.c_int)] + [C.POINTER(vp)] * 5
    L.lsa_preprocess_batch.argtypes = [vp, C.c_int, vp, vp, vp]
    L.lsa_preprocess_stages_batch.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
    L.lsa_preprocess_stages_host.argtypes = [vp, vp, vp, vp, C.c_int]
    L.lsa_psm_stage_batch.argtypes = [vp, C.c_int, C.c_int, C.c_float] + [vp] * 6
    L.lsa_scan_project_host.argtypes = [vp] + [vp] * 6
    L.lsa_orientation_search_host.argtypes = [vp] + [vp] * 6
    L.lsa_translation_estimation_host.argtypes = [vp] + [vp] * 4 + [C.c_float] + [vp] * 4
    L.lsa_match_batch.argtypes = [vp, C.c_int, C.c_int] + [vp] * 12
    L.lsa_error_index2_batch.argtypes = [vp, C.c_int] + [vp] * 8
    L.lsa_scan_project_batch.argtypes = [vp, C.c_int] + [vp] * 6
    L.lsa_fmatch_host.argtypes = [vp] + [vp] * 6 + [C.c_int, vp, vp]
    L.lsa_preprocess_host.argtypes = [vp, vp, vp, vp]
    L.lsa_match_host.argtypes = [vp, C.c_int] + [vp] * 10
    L.lsa_error_index2_host.argtypes = [vp] + [vp] * 9
    L.lsa_grid_create.argtypes = [C.POINTER(vp), vp] + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_double]
    L.lsa_grid_clear.argtypes = [vp]
    L.lsa_grid_update_batch.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp]
    L.lsa_grid_update_host.argtypes = [vp, vp, vp, C.c_int, C.c_float, C.c_float]
    L.lsa_grid_compute_prob.argtypes = [vp]
    L.lsa_grid_download.argtypes = [vp, C.c_int, vp]
    L.lsa_grid_device_ptrs.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.lsa_grid_cloud.argtypes = [vp, C.c_float, vp, vp, C.c_int, C.POINTER(C.c_int)]
    L.lsa_grid_cell.argtypes = [vp, C.c_float, C.c_float, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.lsa_field_create.argtypes = [C.POINTER(vp), vp] + [C.c_int] * 4 + [C.c_float, C.c_float, C.c_double]
    L.lsa_field_destroy.argtypes = [vp]
    L.lsa_field_destroy.restype = None
    L.lsa_field_build.argtypes = [vp, vp, C.c_float, C.c_float, C.c_float]
    L.lsa_field_build_from_grid.argtypes = [vp, vp, C.c_float, C.c_float, C.c_float]
    L.lsa_field_download.argtypes = [vp, C.c_int, vp]
    L.lsa_score_poses.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.lsa_score_poses_std.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.c_int, vp]
    L.lsa_pose_argmax.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int32)]
    L.lsa_pf_weights.argtypes = [vp, vp, C.c_int, C.c_float, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.lsa_pf_resample.argtypes = [vp, vp, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, vp]
    L.lsa_pf_resample_standard.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, vp]
    L.lsa_pose_topk.argtypes = [vp, vp, C.c_int, C.c_int, vp, vp]
    L.lsa_pf_estimate.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, C.POINTER(C.c_double), C.POINTER(C.c_double), vp]
    L.lsa_pf_predict.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.lsa_pf_motion.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.lsa_pf_roughen.argtypes = [vp, C.c_int, C.c_double, vp, vp]
    L.lsa_simulate_scans.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, vp, vp]
    L.lsa_valid_poses.argtypes = [vp, vp, vp, C.c_int, C.c_float, C.c_float, vp]
    L.lsa_beam_model_scores.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, vp]
    L.lsa_plicp_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp]
    L.lsa_plicp_host.argtypes = [vp, vp, vp, vp, vp]
    L.lsa_plicp_batch_cov.argtypes = [vp, C.c_int, vp, vp, vp, vp, C.c_double, vp, vp]
    L.lsa_plicp_host_cov.argtypes = [vp, vp, vp, vp, C.c_double, vp, vp]
    L.lsa_ctx_arena.argtypes = [vp, C.c_uint64, C.POINTER(vp)]
    L.lsa_comm_unique_id.argtypes = [vp]
    L.lsa_comm_create.argtypes = [C.POINTER(vp), vp, C.c_int, C.c_int, vp]
    L.lsa_comm_destroy.argtypes = [vp]
    L.lsa_comm_destroy.restype = None
    L.lsa_comm_rank.argtypes = [vp]
    L.lsa_comm_world.argtypes = [vp]
    L.lsa_comm_bytes_sent.argtypes = [vp]
    L.lsa_comm_bytes_sent.restype = C.c_uint64
    L.lsa_shard_bounds.argtypes = [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    L.lsa_grid_allreduce.argtypes = [vp, vp]
    L.lsa_pf_global_best.argtypes = [vp, vp, vp, C.c_int, C.c_int64, C.POINTER(C.c_float), C.POINTER(C.c_int64), vp]
    L.lsa_pf_global_weights.argtypes = [vp, vp, C.c_int, C.c_float, vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.lsa_pf_resample_sharded.argtypes = [vp, vp, vp, C.c_int, C.c_int64, C.c_double, vp]
    L.lsa_pf_roughen_sharded.argtypes = [vp, C.c_int, C.c_int64, C.c_double, vp, vp]
    L.lsa_pf_fixed_scan.argtypes = [vp, vp, C.c_int, vp, C.POINTER(C.c_uint64)]
    L.lsa_pf_plan.argtypes = [C.c_int, C.c_int64, vp, C.c_double, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), vp]
    L.lsa_pf_offspring_bounds.argtypes = [vp, vp, C.c_int, C.c_uint64, C.c_int64, C.c_uint64, C.c_uint64, vp]
    L.lsa_pf_parents.argtypes = [vp, vp, C.c_int, C.c_int64, C.c_int, vp]
    L.lsa_pf_exchange_plan.argtypes = [C.c_int, C.c_int64, vp, C.c_int, vp, vp]
    L.lsa_pf_filter_create.argtypes = [C.POINTER(vp), vp, C.c_int]
    L.lsa_pf_filter_destroy.argtypes = [vp]
    L.lsa_pf_filter_destroy.restype = None
    L.lsa_pf_filter_slots.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
    L.lsa_pf_filter_score.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, C.c_int64]
    L.lsa_pf_filter_weights.argtypes = [vp, vp, C.c_int, vp, vp]
    L.lsa_pf_filter_read.argtypes = [vp, vp, C.c_int, C.c_int64, C.POINTER(PfInfo), vp]
    L.lsa_pf_filter_offspring.argtypes = [vp, C.c_uint64, C.c_int64, C.c_uint64, C.c_uint64, C.c_int64, C.c_int, vp, vp]
    L.lsa_pf_filter_roughen.argtypes = [vp, vp, C.c_int, C.c_int64, C.c_double, vp, vp]
    L.lsa_pf_filter_step.argtypes = [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_int, C.c_int64,
                                     C.c_double, C.c_double, C.c_double, vp, C.POINTER(PfInfo)]
    L.lsa_icp_goodness_scores.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, C.c_float, C.c_float, C.c_float, vp, vp, vp]
    L.lsa_gather_poses.argtypes = [vp, vp, vp, C.c_int, vp]
    L.lsa_p2p_icp_host.argtypes = [vp, vp, vp, C.c_int, vp, vp, C.c_int, vp, vp, vp]
    L.lsa_p2p_icp_batch.argtypes = [vp, C.c_int, vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp]
    _lib = L
    return L


def check(rc, what=""):
    if rc != 0:
        raise LsaError("%s failed (%d): %s" % (what or "lsa call", rc, lib().lsa_last_error().decode()))


def _hp(a):
    """host numpy array -> void*"""
    return None if a is None else a.ctypes.data_as(vp)


def _dp(t):
    """device pointer: int, None, or anything with .data_ptr() (torch tensor)"""
    if t is None:
        return None
    if isinstance(t, int):
        return vp(t)
    return vp(t.data_ptr())


def param_preset(name):
    p = LaserParam()
    check(lib().lsa_param_preset(C.byref(p), name.encode()), "lsa_param_preset")
    return p


def param_make(points, fov_deg, max_range, min_valid, window, fi_min_deg, fi_max_deg):
    p = LaserParam()
    check(lib().lsa_param_make(C.byref(p), points, fov_deg, max_range, min_valid, window, fi_min_deg, fi_max_deg),
          "lsa_param_make")
    return p


def param_tables(p):
    n = p.points
    fi, si, co = (np.zeros(n, np.float32) for _ in range(3))
    check(lib().lsa_param_tables(C.byref(p), _hp(fi), _hp(si), _hp(co)), "lsa_param_tables")
    return fi, si, co


def carmen_read(p, filename, tag="ROBOTLASER1"):
    """Returns (r, x, y, bad, pose) host arrays; updates p in place like readCarmon."""
    n_scans = C.c_int(0)
    ptrs = [vp() for _ in range(5)]
    check(lib().lsa_carmen_read(C.byref(p), filename.encode(), tag.encode(), C.byref(n_scans),
                                *[C.byref(q) for q in ptrs]), "lsa_carmen_read")
    S, n = n_scans.value, p.points
    outs = []
    for q, (ct, dt, w) in zip(ptrs, [(C.c_float, np.float32, n)] * 3 + [(C.c_int32, np.int32, n), (C.c_float, np.float32, 3)]):
        a = np.ctypeslib.as_array(C.cast(q, C.POINTER(ct)), (S, w)).astype(dt, copy=True)
        lib().lsa_host_free(q)
        outs.append(a)
    return tuple(outs)


class Context:
    """One HIP stream + scratch bound to a laser parameter block."""

    def __init__(self, param, device=0):
        self.L = lib()
        self.h = vp()
        self.param = param
        check(self.L.lsa_ctx_create(C.byref(self.h), device, C.byref(param)), "lsa_ctx_create")

    def close(self):
        if self.h:
            self.L.lsa_ctx_destroy(self.h)
            self.h = vp()

    def __del__(self):
        # (not at interpreter shutdown: the HIP runtime may already be gone, and calling into it then aborts the process)
        if sys is None or sys.is_finalizing():          # (module globals are cleared late in shutdown)
            return
        try:
            self.close()
        except Exception:
            pass

    @property
    def n(self):
        return self.param.points

    def set_param(self, param):
        check(self.L.lsa_ctx_set_param(self.h, C.byref(param)), "lsa_ctx_set_param")
        self.param = param

    def set_stream(self, stream_ptr):
        check(self.L.lsa_ctx_set_stream(self.h, vp(stream_ptr) if stream_ptr else None), "lsa_ctx_set_stream")

    def stream(self):
        return self.L.lsa_ctx_get_stream(self.h)

    def sync(self):
        check(self.L.lsa_ctx_sync(self.h), "lsa_ctx_sync")

    # ---- device-pointer (batched) API: arguments are torch CUDA tensors or raw ints
    def preprocess_batch(self, S, d_r, d_bad, d_seg):
        check(self.L.lsa_preprocess_batch(self.h, S, _dp(d_r), _dp(d_bad), _dp(d_seg)), "lsa_preprocess_batch")

    def match_batch(self, which, B, d_r, d_bad, d_seg, d_ref_idx, d_act_idx, d_pose_out, d_init_pose=None,
                    d_ret=None, d_err_xy=None, d_n_assoc=None, d_status=None, d_iters=None):
        check(self.L.lsa_match_batch(self.h, which, B, _dp(d_r), _dp(d_bad), _dp(d_seg), _dp(d_ref_idx),
                                     _dp(d_act_idx), _dp(d_init_pose), _dp(d_pose_out), _dp(d_ret),
                                     _dp(d_err_xy), _dp(d_n_assoc), _dp(d_status), _dp(d_iters)), "lsa_match_batch")

    def error_index2_batch(self, B, d_r, d_bad, d_seg, d_ref_idx, d_act_idx, d_init_pose, d_err_xy, d_n_assoc=None):
        check(self.L.lsa_error_index2_batch(self.h, B, _dp(d_r), _dp(d_bad), _dp(d_seg), _dp(d_ref_idx),
                                            _dp(d_act_idx), _dp(d_init_pose), _dp(d_err_xy), _dp(d_n_assoc)),
              "lsa_error_index2_batch")

    def scan_project_batch(self, S, d_r, d_bad, d_seg, d_pose, d_new_r, d_new_bad):
        check(self.L.lsa_scan_project_batch(self.h, S, _dp(d_r), _dp(d_bad), _dp(d_seg), _dp(d_pose),
                                            _dp(d_new_r), _dp(d_new_bad)), "lsa_scan_project_batch")

    # ---- single-pair host API (numpy in/out, int flags like struct _PMScan)
    def preprocess_host(self, r, bad=None):
        r = np.ascontiguousarray(r, np.float32).copy()
        bad = np.zeros(self.n, np.int32) if bad is None else np.ascontiguousarray(bad, np.int32).copy()
        seg = np.zeros(self.n, np.int32)
        check(self.L.lsa_preprocess_host(self.h, _hp(r), _hp(bad), _hp(seg)), "lsa_preprocess_host")
        return r, bad, seg

    def preprocess_stages_host(self, r, bad, seg, stages):
        r = np.ascontiguousarray(r, np.float32).copy()
        bad = np.ascontiguousarray(bad, np.int32).copy()
        seg = np.ascontiguousarray(seg, np.int32).copy()
        check(self.L.lsa_preprocess_stages_host(self.h, _hp(r), _hp(bad), _hp(seg), stages), "lsa_preprocess_stages_host")
        return r, bad, seg

    def scan_project_host(self, r, bad, seg, pose):
        pose = np.ascontiguousarray(pose, np.float32)
        new_r = np.zeros(self.n, np.float32)
        new_bad = np.zeros(self.n, np.int32)
        check(self.L.lsa_scan_project_host(self.h, _hp(r), _hp(bad), _hp(seg), _hp(pose), _hp(new_r), _hp(new_bad)),
              "lsa_scan_project_host")
        return new_r, new_bad

    def orientation_search_host(self, ref_r, ref_bad, new_r, new_bad):
        dth = C.c_float(0)
        st = C.c_int32(0)
        check(self.L.lsa_orientation_search_host(self.h, _hp(ref_r), _hp(ref_bad), _hp(new_r), _hp(new_bad),
                                                 C.cast(C.byref(dth), vp), C.cast(C.byref(st), vp)),
              "lsa_orientation_search_host")
        return st.value, dth.value

    def translation_estimation_host(self, ref_r, ref_bad, new_r, new_bad, weight_c=4900.0):
        dx, dy, err = C.c_float(0), C.c_float(0), C.c_float(0)
        st = C.c_int32(0)
        check(self.L.lsa_translation_estimation_host(self.h, _hp(ref_r), _hp(ref_bad), _hp(new_r), _hp(new_bad),
                                                     weight_c, C.cast(C.byref(dx), vp), C.cast(C.byref(dy), vp),
                                                     C.cast(C.byref(err), vp), C.cast(C.byref(st), vp)),
              "lsa_translation_estimation_host")
        return st.value, dx.value, dy.value, err.value

    def match_host(self, which, ref, act, ref_pose=None, act_pose=None):
        rp = None if ref_pose is None else np.ascontiguousarray(ref_pose, np.float32)
        ap = None if act_pose is None else np.ascontiguousarray(act_pose, np.float32)
        pose = np.zeros(3, np.float32)
        ret = C.c_float(0)
        st = C.c_int32(0)
        check(self.L.lsa_match_host(self.h, which, _hp(ref[0]), _hp(ref[1]), _hp(rp),
                                    _hp(act[0]), _hp(act[1]), _hp(act[2]), _hp(ap), _hp(pose),
                                    C.cast(C.byref(ret), vp), C.cast(C.byref(st), vp)), "lsa_match_host")
        return st.value, pose, ret.value

    def error_index2_host(self, ref, act, ref_pose, act_pose):
        rp = np.ascontiguousarray(ref_pose, np.float32)
        ap = np.ascontiguousarray(act_pose, np.float32)
        exy = np.zeros(2, np.float32)
        na = C.c_int32(0)
        check(self.L.lsa_error_index2_host(self.h, _hp(ref[0]), _hp(ref[1]), _hp(rp), _hp(act[0]), _hp(act[1]),
                                           _hp(act[2]), _hp(ap), _hp(exy), C.cast(C.byref(na), vp)),
              "lsa_error_index2_host")
        return exy, na.value

    def fmatch_host(self, ref_raw, act_raw, first):
        ref = [np.ascontiguousarray(a).copy() for a in ref_raw]
        act = [np.ascontiguousarray(a).copy() for a in act_raw]
        pose = np.zeros(3, np.float32)
        pair = np.zeros(2, np.float32)
        check(self.L.lsa_fmatch_host(self.h, _hp(ref[0]), _hp(ref[1]), _hp(ref[2]), _hp(act[0]), _hp(act[1]),
                                     _hp(act[2]), int(bool(first)), _hp(pose), _hp(pair)), "lsa_fmatch_host")
        return pose, pair, ref, act


class Grid:
    """CPMap counting grid resident in HBM."""

    def __init__(self, ctx, sx, sy, cx, cy, offx, offy, res):
        self.L = lib()
        self.ctx = ctx
        self.shape = (sx, sy)
        self.h = vp()
        check(self.L.lsa_grid_create(C.byref(self.h), ctx.h, sx, sy, cx, cy, offx, offy, res), "lsa_grid_create")

    def close(self):
        if self.h:
            self.L.lsa_grid_destroy(self.h)
            self.h = vp()

    def __del__(self):
        # (not at interpreter shutdown: the HIP runtime may already be gone, and calling into it then aborts the process)
        if sys is None or sys.is_finalizing():          # (module globals are cleared late in shutdown)
            return
        try:
            self.close()
        except Exception:
            pass

    def clear(self):
        check(self.L.lsa_grid_clear(self.h), "lsa_grid_clear")

    def update_batch(self, S, n_beams, d_gx, d_gy, d_robot_xy, d_n_valid=None):
        check(self.L.lsa_grid_update_batch(self.h, S, n_beams, _dp(d_gx), _dp(d_gy), _dp(d_robot_xy), _dp(d_n_valid)),
              "lsa_grid_update_batch")

    def update_host(self, gx, gy, rx, ry):
        gx = np.ascontiguousarray(gx, np.float32)
        gy = np.ascontiguousarray(gy, np.float32)
        check(self.L.lsa_grid_update_host(self.h, _hp(gx), _hp(gy), len(gx), rx, ry), "lsa_grid_update_host")

    def compute_prob(self):
        check(self.L.lsa_grid_compute_prob(self.h), "lsa_grid_compute_prob")

    def download(self, layer):
        out = np.zeros(self.shape, np.float32)
        check(self.L.lsa_grid_download(self.h, layer, _hp(out)), "lsa_grid_download")
        return out

    def layers(self):
        return self.download(0), self.download(1), self.download(2)

    def device_ptrs(self):
        a, b, c = vp(), vp(), vp()
        check(self.L.lsa_grid_device_ptrs(self.h, C.byref(a), C.byref(b), C.byref(c)), "lsa_grid_device_ptrs")
        return a.value, b.value, c.value

    def cloud(self, thre, cap=1 << 22):
        x = np.zeros(cap, np.float32)
        y = np.zeros(cap, np.float32)
        n = C.c_int(0)
        check(self.L.lsa_grid_cloud(self.h, thre, _hp(x), _hp(y), cap, C.byref(n)), "lsa_grid_cloud")
        return x[:min(n.value, cap)], y[:min(n.value, cap)]

    def cell(self, px, py):
        ix, iy = C.c_int(0), C.c_int(0)
        check(self.L.lsa_grid_cell(self.h, px, py, C.byref(ix), C.byref(iy)), "lsa_grid_cell")
        return ix.value, iy.value


class Field:
    """Likelihood field of a grid map (CVPmap::obsLikelyhood3) resident in HBM."""

    def __init__(self, ctx, sx, sy, cx, cy, offx, offy, res):
        self.L = lib()
        self.ctx = ctx
        self.shape = (sx, sy)
        self.h = vp()
        check(self.L.lsa_field_create(C.byref(self.h), ctx.h, sx, sy, cx, cy, offx, offy, res), "lsa_field_create")

    def close(self):
        if self.h:
            self.L.lsa_field_destroy(self.h)
            self.h = vp()

    def __del__(self):
        # (not at interpreter shutdown: the HIP runtime may already be gone, and calling into it then aborts the process)
        if sys is None or sys.is_finalizing():          # (module globals are cleared late in shutdown)
            return
        try:
            self.close()
        except Exception:
            pass

    def build(self, d_prob, max_corr=1.0, free_thr=0.3, unknown=0.5):
        check(self.L.lsa_field_build(self.h, _dp(d_prob), max_corr, free_thr, unknown), "lsa_field_build")

    def build_from_grid(self, grid, max_corr=1.0, free_thr=0.3, unknown=0.5):
        check(self.L.lsa_field_build_from_grid(self.h, grid.h, max_corr, free_thr, unknown), "lsa_field_build_from_grid")

    def download(self, what=0):
        out = np.zeros(self.shape, np.float32)
        check(self.L.lsa_field_download(self.h, what, _hp(out)), "lsa_field_download")
        return out

    def score_poses(self, d_scan_x, d_scan_y, n_points, decimation, d_poses, P, d_score, std_frame=False):
        fn = self.L.lsa_score_poses_std if std_frame else self.L.lsa_score_poses
        check(fn(self.h, _dp(d_scan_x), _dp(d_scan_y), n_points, decimation, _dp(d_poses), P, _dp(d_score)), "lsa_score_poses")


def pose_argmax(ctx, d_score, P):
    m, i = C.c_float(0), C.c_int32(0)
    check(lib().lsa_pose_argmax(ctx.h, _dp(d_score), P, C.byref(m), C.byref(i)), "lsa_pose_argmax")
    return m.value, i.value


def pf_weights(ctx, d_score, P, max_score, d_w):
    s, s2 = C.c_double(0), C.c_double(0)
    check(lib().lsa_pf_weights(ctx.h, _dp(d_score), P, max_score, _dp(d_w), C.byref(s), C.byref(s2)), "lsa_pf_weights")
    return s.value, s2.value


# defaults of CVPmap::laserScanSimulator (VPmap.h:12-31)
SIM_FOV = 4.712388981
SIM_MIN_ANGLE = 3 * -0.7853981635
SIM_MAX_RANGE = 30.0


def pose_topk(ctx, d_score, n, k):
    """threadLocalization::topK: indices and scores of the k best, best first."""
    k = min(k, n)
    idx = np.zeros(k, np.int32); sc = np.zeros(k, np.float32)
    check(lib().lsa_pose_topk(ctx.h, _dp(d_score), n, k, _hp(idx), _hp(sc)), "lsa_pose_topk")
    return idx, sc


def pf_estimate(ctx, d_poses, d_score, n, k):
    """CParticles::weightMean + meanDis over the k best: (mean pose[3], mean weight, mean sq. distance, indices)."""
    k = min(k, n)
    mean = np.zeros(3, np.float64); idx = np.zeros(k, np.int32)
    mw, md = C.c_double(0), C.c_double(0)
    check(lib().lsa_pf_estimate(ctx.h, _dp(d_poses), _dp(d_score), n, k, _hp(mean), C.byref(mw), C.byref(md), _hp(idx)),
          "lsa_pf_estimate")
    return mean, mw.value, md.value, idx


def pf_predict(ctx, n, mean_pose, sigma, d_normals, d_poses):
    m = np.ascontiguousarray(mean_pose, np.float64); sg = np.ascontiguousarray(sigma, np.float64)
    check(lib().lsa_pf_predict(ctx.h, n, _hp(m), _hp(sg), _dp(d_normals), _dp(d_poses)), "lsa_pf_predict")


def pf_motion(ctx, n, delta, sigma, d_normals, d_poses):
    m = np.ascontiguousarray(delta, np.float64); sg = np.ascontiguousarray(sigma, np.float64)
    check(lib().lsa_pf_motion(ctx.h, n, _hp(m), _hp(sg), _dp(d_normals), _dp(d_poses)), "lsa_pf_motion")


def pf_roughen(ctx, n, K, d_normals, d_poses):
    check(lib().lsa_pf_roughen(ctx.h, n, K, _dp(d_normals), _dp(d_poses)), "lsa_pf_roughen")


def simulate_scans(field, d_prob, d_poses, P, n_beams, d_ranges, d_bad, fov=SIM_FOV, min_angle=SIM_MIN_ANGLE,
                   max_range=SIM_MAX_RANGE):
    check(lib().lsa_simulate_scans(field.h, _dp(d_prob), _dp(d_poses), P, n_beams, fov, min_angle, max_range,
                                   _dp(d_ranges), _dp(d_bad)), "lsa_simulate_scans")


def valid_poses(field, d_prob, d_poses, P, d_valid, free_thr=0.3, unknown=0.5):
    """CVPmap::isValidPose for P poses: d_valid (uint8) = 1 where the pose's cell is inside the map, free and known."""
    check(lib().lsa_valid_poses(field.h, _dp(d_prob), _dp(d_poses), P, free_thr, unknown, _dp(d_valid)), "lsa_valid_poses")


def beam_model_scores(field, d_prob, d_poses, P, d_obs_r, d_obs_bad, n_beams, d_score, fov=SIM_FOV,
                      min_angle=SIM_MIN_ANGLE, max_range=SIM_MAX_RANGE):
    check(lib().lsa_beam_model_scores(field.h, _dp(d_prob), _dp(d_poses), P, _dp(d_obs_r), _dp(d_obs_bad), n_beams,
                                      fov, min_angle, max_range, _dp(d_score)), "lsa_beam_model_scores")


def icp_goodness_scores(field, d_prob, d_poses, P, d_obs_x, d_obs_y, n_beams, d_weight, d_rel_pose=None, fov=SIM_FOV,
                        min_angle=SIM_MIN_ANGLE, max_range=SIM_MAX_RANGE):
    """CVPmap::obsLikelyhood: ray-cast a virtual scan per pose and align it with the observed scan by ICP."""
    check(lib().lsa_icp_goodness_scores(field.h, _dp(d_prob), _dp(d_poses), P, _dp(d_obs_x), _dp(d_obs_y), n_beams,
                                        fov, min_angle, max_range, None, _dp(d_weight), _dp(d_rel_pose)),
          "lsa_icp_goodness_scores")


def pf_resample(ctx, d_w, P, u0, n_total, first, M, d_parent):
    check(lib().lsa_pf_resample(ctx.h, _dp(d_w), P, u0, n_total, first, M, _dp(d_parent)), "lsa_pf_resample")


def pf_resample_standard(ctx, d_w, P, d_u, n_u, d_parent, first=0, M=None):
    """Standard_resampler (SIRFlt.cpp:39-115): d_u = n_u uniforms (float64, device); d_parent = offspring
    first .. first + M - 1 of the parent-ascending list (all of it by default)."""
    check(lib().lsa_pf_resample_standard(ctx.h, _dp(d_w), P, _dp(d_u), n_u, first, n_u if M is None else M, _dp(d_parent)),
          "lsa_pf_resample_standard")


def gather_poses(ctx, d_src, d_parent, M, d_dst):
    check(lib().lsa_gather_poses(ctx.h, _dp(d_src), _dp(d_parent), M, _dp(d_dst)), "lsa_gather_poses")


def plicp_batch(ctx, B, d_r, d_ref_idx, d_act_idx, d_out8):
    """CCanonicalMatcher's point-to-line ICP for B pairs of the device scan set d_r (cm); d_out8 [B][8] doubles."""
    check(lib().lsa_plicp_batch(ctx.h, B, _dp(d_r), _dp(d_ref_idx), _dp(d_act_idx), None, _dp(d_out8)), "lsa_plicp_batch")


def plicp_batch_cov(ctx, B, d_r, d_ref_idx, d_act_idx, sigma, d_out8, d_cov9):
    """lsa_plicp_batch_cov: the batch above plus sm_icp's covariance per pair (d_cov9 [B][9] doubles)."""
    check(lib().lsa_plicp_batch_cov(ctx.h, B, _dp(d_r), _dp(d_ref_idx), _dp(d_act_idx), None, float(sigma), _dp(d_out8),
                                    _dp(d_cov9)), "lsa_plicp_batch_cov")


def plicp_host_cov(ctx, ref_r, cur_r, sigma=0.04):
    a = np.ascontiguousarray(ref_r, np.float32); b = np.ascontiguousarray(cur_r, np.float32)
    out = np.zeros(8, np.float64); cov = np.zeros(9, np.float64)
    check(lib().lsa_plicp_host_cov(ctx.h, _hp(a), _hp(b), None, float(sigma), _hp(out), _hp(cov)), "lsa_plicp_host_cov")
    return out, cov.reshape(3, 3)


def plicp_host(ctx, ref_r, cur_r):
    a, b = np.ascontiguousarray(ref_r, np.float32), np.ascontiguousarray(cur_r, np.float32)
    out = np.zeros(8, np.float64)
    check(lib().lsa_plicp_host(ctx.h, _hp(a), _hp(b), None, _hp(out)), "lsa_plicp_host")
    return out


def p2p_icp_batch(ctx, B, d_ref_x, d_ref_y, d_ref_off, d_cur_x, d_cur_y, d_cur_off, n_ref_total, n_cur_total, d_init_pose, d_out16):
    """CICPWarpper::ICPMatch for B problems: clouds concatenated (metres, device), offsets int32 [B + 1], init poses
    float64 [B][3], d_out16 float64 [B][16]; returns after the batch has finished."""
    check(lib().lsa_p2p_icp_batch(ctx.h, int(B), _dp(d_ref_x), _dp(d_ref_y), _dp(d_ref_off), _dp(d_cur_x), _dp(d_cur_y), _dp(d_cur_off),
                                  int(n_ref_total), int(n_cur_total), _dp(d_init_pose), None, _dp(d_out16)), "lsa_p2p_icp_batch")


def p2p_icp_host(ctx, ref_x, ref_y, cur_x, cur_y, init_pose):
    """CICPWarpper::ICPMatch with the reference's options; returns the 16-double result record."""
    rx, ry = np.ascontiguousarray(ref_x, np.float32), np.ascontiguousarray(ref_y, np.float32)
    cx, cy = np.ascontiguousarray(cur_x, np.float32), np.ascontiguousarray(cur_y, np.float32)
    init = np.ascontiguousarray(init_pose, np.float64)
    out = np.zeros(16, np.float64)
    check(lib().lsa_p2p_icp_host(ctx.h, _hp(rx), _hp(ry), len(rx), _hp(cx), _hp(cy), len(cx), _hp(init), None, _hp(out)),
          "lsa_p2p_icp_host")
    return out


# ---- multi-GPU entry points (include/lsa_core.h, "multi-GPU") -------------------------------------------------
def shard_bounds(n_total, rank, world):
    f, c = C.c_int64(0), C.c_int64(0)
    check(lib().lsa_shard_bounds(int(n_total), int(rank), int(world), C.byref(f), C.byref(c)), "lsa_shard_bounds")
    return f.value, c.value


def pf_plan(shard_sums, n_total, u0):
    """lsa_pf_plan (host only, needs no GPU): total, U0 and bounds[world + 1] of the sharded systematic resampling."""
    sums = np.ascontiguousarray(shard_sums, np.uint64)
    world = len(sums)
    W, U0 = C.c_uint64(0), C.c_uint64(0)
    bounds = np.zeros(world + 1, np.int64)
    check(lib().lsa_pf_plan(world, int(n_total), _hp(sums), float(u0), C.byref(W), C.byref(U0), _hp(bounds)), "lsa_pf_plan")
    return W.value, U0.value, bounds


def pf_fixed_scan(ctx, d_w, n, d_cum):
    """d_cum: int64 / uint64 [n] device tensor; returns the shard's total."""
    s = C.c_uint64(0)
    check(lib().lsa_pf_fixed_scan(ctx.h, _dp(d_w), n, _dp(d_cum), C.byref(s)), "lsa_pf_fixed_scan")
    return s.value


def pf_offspring_bounds(ctx, d_cum, n, offset, n_total, U0, W, d_e):
    check(lib().lsa_pf_offspring_bounds(ctx.h, _dp(d_cum), n, int(offset), int(n_total), int(U0), int(W), _dp(d_e)), "lsa_pf_offspring_bounds")


def pf_parents(ctx, d_e, n, m_first, m_count, d_parent):
    check(lib().lsa_pf_parents(ctx.h, _dp(d_e), n, int(m_first), int(m_count), _dp(d_parent)), "lsa_pf_parents")


def pf_exchange_plan(bounds, n_total, rank, world):
    """lsa_pf_exchange_plan (host only): (send[world], recv[world]) of the surplus exchange."""
    b = np.ascontiguousarray(bounds, np.int64)
    send = np.zeros(world, np.int64); recv = np.zeros(world, np.int64)
    check(lib().lsa_pf_exchange_plan(int(world), int(n_total), _hp(b), int(rank), _hp(send), _hp(recv)), "lsa_pf_exchange_plan")
    return [int(v) for v in send], [int(v) for v in recv]


class PfFilter:
    """lsa_pf_filter: the fused localisation step's device buffers and launches for up to `capacity` particles."""

    def __init__(self, field, capacity):
        self.h = vp()
        self.L = lib()
        self.field = field
        check(self.L.lsa_pf_filter_create(C.byref(self.h), field.h, int(capacity)), "lsa_pf_filter_create")
        k, r, m, sc = vp(), vp(), vp(), vp()
        check(self.L.lsa_pf_filter_slots(self.h, C.byref(k), C.byref(r), C.byref(m), C.byref(sc)), "lsa_pf_filter_slots")
        self.d_key, self.d_record, self.d_minmax, self.d_score = k.value, r.value, m.value, sc.value

    def close(self):
        if self.h:
            self.L.lsa_pf_filter_destroy(self.h)
            self.h = vp()

    def score(self, d_scan_x, d_scan_y, n_points, decimation, d_poses, n_local, first_index=0, std_frame=False, predict=None,
              d_normals=None):
        """predict = None | ("predict", mean[3], sigma[3]) | ("motion", delta[3], sigma[3])"""
        mode, m, sg = 0, None, None
        if predict is not None:
            mode = 1 if predict[0] == "predict" else 2
            m = np.ascontiguousarray(predict[1], np.float64); sg = np.ascontiguousarray(predict[2], np.float64)
        check(self.L.lsa_pf_filter_score(self.h, _dp(d_scan_x), _dp(d_scan_y), n_points, decimation, int(std_frame), mode, _hp(m), _hp(sg),
                                         _dp(d_normals), _dp(d_poses), int(n_local), int(first_index)), "lsa_pf_filter_score")

    def weights(self, d_poses, d_keys=None, n_keys=1, d_w=None):
        check(self.L.lsa_pf_filter_weights(self.h, _dp(d_keys), int(n_keys), _dp(d_poses), _dp(d_w)), "lsa_pf_filter_weights")

    def read(self, n_total, d_records=None, world=1):
        info = PfInfo(); sums = np.zeros(world, np.uint64)
        check(self.L.lsa_pf_filter_read(self.h, _dp(d_records), int(world), int(n_total), C.byref(info), _hp(sums)), "lsa_pf_filter_read")
        return info.as_dict(), [int(v) for v in sums]

    def offspring(self, offset, n_total, U0, W, m_first, m_count, d_poses, d_out):
        check(self.L.lsa_pf_filter_offspring(self.h, int(offset), int(n_total), int(U0), int(W), int(m_first), int(m_count), _dp(d_poses),
                                             _dp(d_out)), "lsa_pf_filter_offspring")

    def roughen(self, n, n_total, K, d_normals, d_poses, d_minmax=None):
        check(self.L.lsa_pf_filter_roughen(self.h, _dp(d_minmax), int(n), int(n_total), float(K), _dp(d_normals), _dp(d_poses)),
              "lsa_pf_filter_roughen")

    def step(self, d_scan_x, d_scan_y, n_points, decimation, d_poses, d_poses_out, n_local, n_total, u0, resample_threshold=0.5,
             roughen_K=0.0, d_normals_roughen=None, predict=None, d_normals_predict=None, std_frame=False, comm=None):
        """lsa_pf_filter_step: the whole step in one call (comm = None: one rank)."""
        mode, m, sg = 0, None, None
        if predict is not None:
            mode = 1 if predict[0] == "predict" else 2
            m = np.ascontiguousarray(predict[1], np.float64); sg = np.ascontiguousarray(predict[2], np.float64)
        info = PfInfo()
        check(self.L.lsa_pf_filter_step(self.h, comm.h if comm is not None else None, _dp(d_scan_x), _dp(d_scan_y), n_points, decimation,
                                        int(std_frame), mode, _hp(m), _hp(sg), _dp(d_normals_predict), _dp(d_poses), _dp(d_poses_out),
                                        int(n_local), int(n_total), float(resample_threshold), float(u0), float(roughen_K),
                                        _dp(d_normals_roughen), C.byref(info)), "lsa_pf_filter_step")
        return info.as_dict()


def comm_unique_id():
    buf = np.zeros(128, np.uint8)
    check(lib().lsa_comm_unique_id(_hp(buf)), "lsa_comm_unique_id")
    return buf


class Comm:
    """lsa_comm: an RCCL communicator bound to a context (one rank per GPU); world 1 needs no id and no RCCL."""

    def __init__(self, ctx, rank=0, world=1, unique_id=None):
        self.h = vp()
        self.ctx = ctx
        idp = None if unique_id is None else _hp(np.ascontiguousarray(unique_id, np.uint8))
        check(lib().lsa_comm_create(C.byref(self.h), ctx.h, rank, world, idp), "lsa_comm_create")
        self.rank, self.world = rank, world

    def close(self):
        if self.h:
            lib().lsa_comm_destroy(self.h)
            self.h = vp()

    def bytes_sent(self):
        return int(lib().lsa_comm_bytes_sent(self.h))

    def grid_allreduce(self, grid):
        check(lib().lsa_grid_allreduce(grid.h, self.h), "lsa_grid_allreduce")

    def pf_global_best(self, d_score, d_poses, n_local, first_index):
        sc, idx = C.c_float(0), C.c_int64(0)
        pose = np.zeros(3, np.float32)
        check(lib().lsa_pf_global_best(self.h, _dp(d_score), _dp(d_poses), n_local, int(first_index), C.byref(sc), C.byref(idx),
                                       _hp(pose)), "lsa_pf_global_best")
        return sc.value, idx.value, pose

    def pf_global_weights(self, d_score, n_local, max_score, d_w):
        s, s2 = C.c_double(0), C.c_double(0)
        check(lib().lsa_pf_global_weights(self.h, _dp(d_score), n_local, max_score, _dp(d_w), C.byref(s), C.byref(s2)),
              "lsa_pf_global_weights")
        return s.value, s2.value

    def pf_resample_sharded(self, d_w, d_poses, n_local, n_total, u0, d_poses_out):
        check(lib().lsa_pf_resample_sharded(self.h, _dp(d_w), _dp(d_poses), n_local, int(n_total), float(u0), _dp(d_poses_out)),
              "lsa_pf_resample_sharded")

    def pf_roughen_sharded(self, n_local, n_total, K, d_normals, d_poses):
        check(lib().lsa_pf_roughen_sharded(self.h, n_local, int(n_total), float(K), _dp(d_normals), _dp(d_poses)), "lsa_pf_roughen_sharded")
