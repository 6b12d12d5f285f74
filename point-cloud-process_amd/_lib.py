"""ctypes binding of libpcr.so (the C ABI declared in include/pcr.h).

There is no CPU fallback: if the HIP library is missing or no AMD GPU is
present, every compute entry point raises.  Only symbol loading works on a
GPU-less host (that is what the ``-m "not gpu"`` tests check).
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PCR_LIB_PATH") or os.path.join(_HERE, "libpcr.so")  # override: A/B builds only

PCR_OK = 0
PCR_E_TOO_FEW_ASSOC = 1
PCR_E_INVALID = -1
PCR_E_EMPTY = -2
PCR_E_NOMEM = -3
PCR_E_HIP = -4
PCR_E_NO_DEVICE = -5
PCR_E_UNSUPPORTED = -6
PCR_E_TOO_MANY_ITERS = -7

PCR_INDEX_GRID = 0
PCR_INDEX_BRUTE = 1
PCR_ICP_COMPAT_MAIN = 0
PCR_ICP_TOTAL = 1
PCR_RMETRIC_FROBENIUS = 0
PCR_RMETRIC_GEODESIC = 1
PCR_ICP_MAX_LOG = 256


class IcpParams(C.Structure):
    _fields_ = [
        ("max_iter", C.c_int32),
        ("r_thres", C.c_double),
        ("t_thres", C.c_double),
        ("max_d2", C.c_double),
        ("mode", C.c_int32),
        ("r_metric", C.c_int32),
        ("min_iter", C.c_int32),
        ("reserved", C.c_int32),
    ]


class RansacParams(C.Structure):
    _fields_ = [
        ("max_iteration", C.c_int32),
        ("check_distance", C.c_int32),
        ("confidence", C.c_double),
        ("max_distance", C.c_double),
        ("edge_similarity", C.c_double),
        ("seed", C.c_uint64),
        ("reserved", C.c_double * 4),
    ]


class RansacResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16),
        ("iterations", C.c_int32),
        ("n_valid", C.c_int32),
        ("best_iteration", C.c_int32),
        ("reserved_i", C.c_int32),
        ("corr_fitness", C.c_double),
        ("corr_rmse", C.c_double),
        ("reserved", C.c_double * 4),
    ]


class Pair(C.Structure):
    _fields_ = [
        ("src", C.POINTER(C.c_float)),
        ("n_src", C.c_int64),
        ("stride_src", C.c_int64),
        ("tgt", C.POINTER(C.c_float)),
        ("n_tgt", C.c_int64),
        ("stride_tgt", C.c_int64),
        ("T0", C.POINTER(C.c_double)),
    ]


class CloudRef(C.Structure):
    _fields_ = [("xyz", C.POINTER(C.c_float)), ("n", C.c_int64), ("stride", C.c_int64)]


class PairRef(C.Structure):
    _fields_ = [("src", C.c_int32), ("tgt", C.c_int32), ("T0", C.POINTER(C.c_double))]


class GlobalParams(C.Structure):
    _fields_ = [
        ("voxel_size", C.c_double),
        ("normal_radius", C.c_double),
        ("fpfh_radius", C.c_double),
        ("normal_max_nn", C.c_int32),
        ("fpfh_max_nn", C.c_int32),
        ("mutual_filter", C.c_int32),
        ("reserved_i", C.c_int32),
        ("ransac", RansacParams),
    ]


class IcpResult(C.Structure):
    _fields_ = [
        ("T", C.c_double * 16),
        ("T_total", C.c_double * 16),
        ("iters", C.c_int32),
        ("status", C.c_int32),
        ("n_assoc", C.c_int64),
        ("cost", C.c_double),
        ("mean_d2", C.c_double),
        ("r_diff", C.c_double * PCR_ICP_MAX_LOG),
        ("t_diff", C.c_double * PCR_ICP_MAX_LOG),
        ("device_ms", C.c_double),
        ("nn_kernel_ms", C.c_double),
        ("nn_launches", C.c_int32),
        ("reserved", C.c_int32),
    ]


_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)

# name -> (restype, argtypes); one entry per PCR_API symbol of include/pcr.h
SIGNATURES = {
    "pcr_strerror": (C.c_char_p, [C.c_int]),
    "pcr_last_error": (C.c_char_p, [_vp]),
    "pcr_version": (C.c_char_p, []),
    "pcr_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "pcr_ctx_destroy": (C.c_int, [_vp]),
    "pcr_ctx_sync": (C.c_int, [_vp]),
    "pcr_ctx_device_info": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_int), _lp]),
    "pcr_cloud_upload_f32": (C.c_int, [_vp, _fp, C.c_int64, C.c_int64, C.POINTER(_vp)]),
    "pcr_cloud_upload_f64": (C.c_int, [_vp, _dp, C.c_int64, C.c_int64, C.POINTER(_vp)]),
    "pcr_cloud_download_f64": (C.c_int, [_vp, _vp, _dp]),
    "pcr_cloud_size": (C.c_int64, [_vp]),
    "pcr_cloud_free": (C.c_int, [_vp, _vp]),
    "pcr_cloud_transform": (C.c_int, [_vp, _vp, _dp]),
    "pcr_cloud_prepare": (C.c_int, [_vp, _vp, _vp]),
    "pcr_index_build": (C.c_int, [_vp, _vp, C.c_int, C.c_double, C.POINTER(_vp)]),
    "pcr_index_free": (C.c_int, [_vp, _vp]),
    "pcr_index_kind": (C.c_int, [_vp]),
    "pcr_index_cell": (C.c_double, [_vp]),
    "pcr_index_size": (C.c_int64, [_vp]),
    "pcr_nn1": (C.c_int, [_vp, _vp, _vp, _dp, C.c_double, _ip, _dp]),
    "pcr_knn": (C.c_int, [_vp, _vp, _dp, C.c_int64, C.c_int, _ip, _dp]),
    "pcr_radius": (C.c_int, [_vp, _vp, _dp, C.c_int64, C.c_double, _lp, _lp, _ip, _dp]),
    "pcr_radius_small": (C.c_int, [_vp, _vp, _dp, C.c_int, C.c_double, C.c_int64, _lp, _ip, _dp]),
    "pcr_icp_default_params": (None, [C.POINTER(IcpParams)]),
    "pcr_icp": (C.c_int, [_vp, _vp, _vp, C.POINTER(IcpParams), _dp, C.POINTER(IcpResult)]),
    "pcr_icp_batch": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(Pair), C.c_int64, C.POINTER(IcpParams), C.POINTER(IcpResult), _ip]),
    "pcr_global_default_params": (C.c_int, [C.c_double, C.POINTER(GlobalParams)]),
    "pcr_register_pairs": (C.c_int, [C.POINTER(_vp), C.c_int, C.POINTER(CloudRef), C.c_int64, C.POINTER(PairRef), C.c_int64, C.POINTER(GlobalParams),
                                     C.POINTER(IcpParams), C.POINTER(IcpResult), _ip, _dp]),
    "pcr_icp_moments": (C.c_int, [_vp, _vp, _vp, _dp, C.c_double, _dp, _dp, _dp]),
    "pcr_procrustes": (C.c_int, [_dp, _dp, C.c_int64, _dp, _dp, _dp]),
    "pcr_homo2tq": (C.c_int, [_dp, _dp]),
    "pcr_voxel_keys": (C.c_int, [_vp, _dp, C.c_int64, C.c_double, _dp, _dp]),
    "pcr_voxel_filter": (C.c_int, [_vp, _dp, C.c_int64, C.c_double, C.c_int, C.c_uint64, _dp, _lp]),
    "pcr_voxel_filter_cloud": (C.c_int, [_vp, _vp, C.c_double, C.c_int, C.c_uint64, C.POINTER(_vp)]),
    "pcr_iss": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, _dp, _ip, _ip, C.POINTER(C.c_int)]),
    "pcr_pca": (C.c_int, [_vp, _vp, _dp, _dp, _dp]),
    "pcr_normals": (C.c_int, [_vp, _vp, C.c_int, _dp, _dp, _ip]),
    "pcr_normals_hybrid": (C.c_int, [_vp, _vp, C.c_double, C.c_int, C.c_int, _dp, _dp]),
    "pcr_fpfh": (C.c_int, [_vp, _vp, _dp, C.c_double, C.c_int, _dp]),
    "pcr_feature_match": (C.c_int, [_vp, _dp, C.c_int64, _dp, C.c_int64, C.c_int, _ip, _dp]),
    "pcr_ransac_default_params": (C.c_int, [C.POINTER(RansacParams)]),
    "pcr_ransac": (C.c_int, [_vp, _vp, _vp, _ip, C.c_int64, C.POINTER(RansacParams), C.POINTER(RansacResult)]),
    "pcr_preprocess": (C.c_int, [_vp, _vp, C.c_double, C.c_double, C.c_int, C.c_double, C.c_int, C.POINTER(_vp)]),
    "pcr_prep_size": (C.c_int64, [_vp]),
    "pcr_prep_cloud": (_vp, [_vp]),
    "pcr_prep_download": (C.c_int, [_vp, _vp, _dp, _dp, _dp]),
    "pcr_prep_free": (C.c_int, [_vp, _vp]),
    "pcr_global_registration": (C.c_int, [_vp, _vp, _vp, C.POINTER(RansacParams), C.c_int, C.POINTER(RansacResult)]),
    "pcr_dbscan": (C.c_int, [_vp, _vp, C.c_double, C.c_int, _ip, _ip]),
    "pcr_debug_read": (C.c_int, [_vp, C.POINTER(C.c_uint64), C.c_int64]),
    "pcr_profile_enable": (C.c_int, [_vp, C.c_int]),
    "pcr_profile_read": (C.c_int, [_vp, _dp, C.POINTER(C.c_int)]),
    "pcr_search_stats": (C.c_int, [_vp, _lp]),
    "pcr_ctx_set_shared": (C.c_int, [_vp, C.c_int]),
    "pcr_icp_pass_log": (C.c_int, [_vp, C.c_int, _dp, _dp, _lp, C.POINTER(C.c_int), _dp]),
    "pcr_timer_start": (C.c_int, [_vp]),
    "pcr_timer_stop_ms": (C.c_int, [_vp, _dp]),
}

_lib = None


class PcrError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = f"libpcr status {status}"
        try:
            msg += ": " + lib().pcr_strerror(status).decode()
        except Exception:
            pass
        if detail:
            msg += f" ({detail})"
        super().__init__(msg)


def lib():
    """Load libpcr.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
            "There is no CPU fallback for the registration path."
        )
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (same soname);
    # importing torch first makes libpcr.so bind to that copy instead of loading a
    # second runtime from /opt/rocm.
    if "torch" not in sys.modules and os.environ.get("PCR_NO_TORCH_PRELOAD", "0") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status, ctx=None, soft=(PCR_E_TOO_FEW_ASSOC,)):
    if status == PCR_OK or status in soft:
        return status
    detail = ""
    if ctx is not None and status == PCR_E_HIP:
        try:
            detail = lib().pcr_last_error(ctx).decode()
        except Exception:
            pass
    raise PcrError(status, detail)


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def lptr(a):
    return a.ctypes.data_as(_lp)
