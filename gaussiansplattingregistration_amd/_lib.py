"""Loader of the C-ABI HIP library (csrc/libgsr_hip.so, declared in include/gsr_hip.h).

The product path has NO CPU fallback: if the library is missing, fails to load, or no HIP device
is visible, every entry point raises ``RuntimeError``.  Build it with ``__graft_entry__.build()``
(hipcc cross-compiles for gfx950 without a GPU).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GSR_HIP_LIB") or os.path.join(_HERE, "csrc", "libgsr_hip.so")   # GSR_HIP_LIB: A/B builds

GSR_OK = 0
GSR_E_INVALID = -1
GSR_E_HIP = -2
GSR_E_NO_DEVICE = -3
GSR_E_PRECONDITION = -4

GSR_RNG_GLIBC = 0
GSR_RNG_HASH = 1

GSR_ICP_ACC_LEN = 32

GSR_COMM_ID_BYTES = 128

GSR_DECOMP_REFERENCE = 0
GSR_DECOMP_EXACT = 1

ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.POINTER(C.c_double), C.c_int32, C.c_void_p)
ALLREDUCE_DEV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int64, C.c_void_p)
ALLGATHER_DEV_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)

COMM_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p)
COMM_ALLGATHER_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
COMM_EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                               C.POINTER(C.c_int64), C.c_void_p)


class CommCallbacks(C.Structure):          # struct gsr_comm_callbacks
    _fields_ = [("allreduce", COMM_ALLREDUCE_FN), ("allgather", COMM_ALLGATHER_FN), ("exchange", COMM_EXCHANGE_FN), ("user", C.c_void_p)]


# name -> (restype, argtypes); mirrors include/gsr_hip.h one to one (tests/test_abi.py checks it)
_vp, _i32, _i64, _u32, _u64, _f32, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float, C.c_double
SIGNATURES = {
    "gsr_last_error": (C.c_char_p, []),
    "gsr_version": (C.c_char_p, []),
    "gsr_device_count": (_i32, []),
    "gsr_comm_get_unique_id": (_i32, [_vp]),
    "gsr_comm_create": (_i32, [C.POINTER(_vp), _vp, _i32, _i32, _i32]),
    "gsr_comm_create_callbacks": (_i32, [C.POINTER(_vp), _i32, _i32, _i32, C.POINTER(CommCallbacks)]),
    "gsr_comm_destroy": (_i32, [_vp]),
    "gsr_comm_rank": (_i32, [_vp]),
    "gsr_comm_world": (_i32, [_vp]),
    "gsr_comm_allreduce": (_i32, [_vp, _vp, _i64, _i32, _i32, _vp]),
    "gsr_comm_allgather": (_i32, [_vp, _vp, _vp, _i64, _vp]),
    "gsr_comm_exchange": (_i32, [_vp, _vp, C.POINTER(_i64), C.POINTER(_i64), _vp, C.POINTER(_i64), C.POINTER(_i64), _vp]),
    "gsr_hem_create": (_i32, [C.POINTER(_vp), _i32, _vp]),
    "gsr_hem_destroy": (_i32, [_vp]),
    "gsr_hem_set_params": (_i32, [_vp, _f32, _f32, _f32, _f32]),
    "gsr_hem_set_rng": (_i32, [_vp, _i32, _u32, _u64]),
    "gsr_hem_get_rng_position": (_i32, [_vp, C.POINTER(_u64)]),
    "gsr_hem_set_shard": (_i32, [_vp, _i32, _i32, ALLREDUCE_DEV_FN, ALLGATHER_DEV_FN, _vp]),
    "gsr_hem_set_level0": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i32]),
    "gsr_hem_set_state": (_i32, [_vp, _vp, _vp]),
    "gsr_hem_set_comm": (_i32, [_vp, _vp]),
    "gsr_hem_set_level0_part": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _i32]),
    "gsr_hem_get_gids": (_i32, [_vp, _vp, _i32]),
    "gsr_hem_set_output": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _i64]),
    "gsr_hem_get_part_stats": (_i32, [_vp, C.POINTER(_i64)]),
    "gsr_hem_get_part_ms": (_i32, [_vp, C.POINTER(C.c_float)]),
    "gsr_hem_run_level": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "gsr_hem_run_levels": (_i32, [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    "gsr_hem_level_size": (_i32, [_vp, C.POINTER(_i64), C.POINTER(_i32)]),
    "gsr_hem_get_level": (_i32, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32]),
    "gsr_hem_get_stats": (_i32, [_vp, C.POINTER(_i64)]),
    "gsr_hem_get_stats_ex": (_i32, [_vp, C.POINTER(_i64)]),
    "gsr_hem_get_phase_ms": (_i32, [_vp, C.POINTER(_f32)]),
    "gsr_hem_get_kernel_ms": (_i32, [_vp, C.POINTER(_f32)]),
    "gsr_hem_set_timing": (_i32, [_vp, _i32]),
    "gsr_icp_create": (_i32, [C.POINTER(_vp), _i32, _vp]),
    "gsr_icp_destroy": (_i32, [_vp]),
    "gsr_icp_set_target": (_i32, [_vp, _vp, _vp, _i64, _f64, _i32]),
    "gsr_icp_set_source": (_i32, [_vp, _vp, _i64, _i32]),
    "gsr_icp_set_target_cov": (_i32, [_vp, _vp, _i32]),
    "gsr_voxel_down_sample": (_i32, [_i32, _vp, _vp, _vp, _vp, _i64, _f64, _i32, C.POINTER(_vp), C.POINTER(_i64)]),
    "gsr_voxel_fetch": (_i32, [_vp, _vp, _vp, _vp, _i32]),
    "gsr_voxel_free": (_i32, [_vp]),
    "gsr_icp_set_source_cov": (_i32, [_vp, _vp, _i32]),
    "gsr_icp_set_target_color": (_i32, [_vp, _vp, _i32]),
    "gsr_icp_set_source_color": (_i32, [_vp, _vp, _i32]),
    "gsr_icp_set_lambda_geometric": (_i32, [_vp, _f64]),
    "gsr_icp_get_color_gradient": (_i32, [_vp, _vp]),
    "gsr_icp_set_allreduce": (_i32, [_vp, ALLREDUCE_FN, _vp, _i64]),
    "gsr_icp_set_allreduce_dev": (_i32, [_vp, ALLREDUCE_DEV_FN, _vp, _i64]),
    "gsr_icp_set_comm": (_i32, [_vp, _vp, _i64]),
    "gsr_icp_accumulate": (_i32, [_vp, _vp, _i32, _i32, _f64, _vp]),
    "gsr_icp_register_clouds": (_i32, [_vp, _vp, _i64, _vp, _vp, _i64, _i32, _f64, _vp, _i32, _i32, _f64, _f64, _f64, _i32, _vp, C.POINTER(_f64),
                                       C.POINTER(_f64), C.POINTER(_i32)]),
    "gsr_icp_register_multiscale": (_i32, [_vp, _i32, _vp, _i32, _vp, _i32, _i32, _f64, _f64, _f64, _vp, _vp]),
    "gsr_icp_register": (_i32, [_vp, _vp, _i32, _i32, _f64, _f64, _f64, _i32, _vp, C.POINTER(_f64), C.POINTER(_f64),
                                C.POINTER(_i32)]),
    "gsr_icp_correspondences": (_i32, [_vp, _vp, _vp, _vp]),
    "gsr_icp_get_timing": (_i32, [_vp, C.POINTER(_f32)]),
    "gsr_normals_from_cov": (_i32, [_vp, _i64, _vp, _i32, _i32, _vp]),
    "gsr_normals_knn": (_i32, [_vp, _i64, _i32, _vp, _i32, _i32, _vp]),
    "gsr_cov_from_normals": (_i32, [_vp, _i64, _f64, _vp, _i32, _i32, _vp]),
    "gsr_decompose_cov": (_i32, [_vp, _i64, _i32, _vp, _vp, _vp, _i32, _i32, _vp]),
    "gsr_ply_unpack": (_i32, [_vp, _i64, _i32, C.POINTER(_i32), _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "gsr_plane_score": (_i32, [_vp, _vp, _i64, _vp, _i32, _f32, _f32, _vp, _vp, C.POINTER(_i32), _i32, _i32, _vp]),
    "gsr_icp_solve": (_i32, [_vp, _i32, _vp, _vp]),
    "gsr_icp_get_centre": (_i32, [_vp, _vp]),
}
# private test hooks (csrc/gsr_test_hooks.h): exported by the library, not part of the public header
TEST_HOOKS = {
    "gsr_debug_logf": (_i32, [_vp, _i64, _vp, _i32]),
    "gsr_debug_kld": (_i32, [_vp, _vp, _vp, _vp, _i64, _vp, _i32]),
    "gsr_debug_kl_gate": (_i32, [_vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _i32]),
    "gsr_debug_stage1": (_i32, [_vp, _vp, _vp, _vp, _i64, _f32, _vp, _vp, _vp, _vp, _vp, _vp, _i32]),
}

_lib = None


class HemLevelReport(C.Structure):
    """gsr_hem_level_report (include/gsr_hip.h): what gsr_hem_run_levels reports per level."""
    _fields_ = [("offset_rows", C.c_int64), ("rows", C.c_int64), ("dropped", C.c_int64), ("rng_position", C.c_uint64),
                ("stats", C.c_int64 * 8), ("stats_ex", C.c_int64 * 8), ("phase_ms", C.c_float * 8), ("kernel_ms", C.c_float * 8)]


class IcpEntry(C.Structure):
    """gsr_icp_entry (include/gsr_hip.h): one entry of a coarse-to-fine schedule."""
    _fields_ = [("src_xyz", C.c_void_p), ("ns", C.c_int64), ("tgt_xyz", C.c_void_p), ("tgt_normals", C.c_void_p), ("nt", C.c_int64),
                ("max_corr", C.c_double), ("max_iter", C.c_int32), ("reserved", C.c_int32)]


class IcpEntryResult(C.Structure):
    """gsr_icp_entry_result (include/gsr_hip.h)."""
    _fields_ = [("init_T", C.c_double * 16), ("T", C.c_double * 16), ("fitness", C.c_double), ("inlier_rmse", C.c_double),
                ("iterations", C.c_int32), ("evaluations", C.c_int32), ("ms_build", C.c_float), ("ms_iters", C.c_float)]


def load(require_device: bool = False):
    """Return the ctypes handle of libgsr_hip.so; raise RuntimeError if it cannot serve the hot path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"HIP extension missing: {LIB_PATH} not built.  Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(needs hipcc).  This backend has no CPU fallback.")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover - depends on the machine
            raise RuntimeError(f"HIP extension failed to load ({LIB_PATH}): {e}.  This backend has no CPU fallback.") from e
        for name, (res, args) in list(SIGNATURES.items()) + list(TEST_HOOKS.items()):
            fn = getattr(lib, name)          # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    if require_device and _lib.gsr_device_count() <= 0:
        raise RuntimeError("no HIP device visible: the MI355X backend has no CPU fallback")
    return _lib


def check(code: int, what: str = ""):
    if code != GSR_OK:
        msg = load().gsr_last_error()
        msg = msg.decode("utf-8", "replace") if msg else ""
        raise RuntimeError(f"{what or 'gsr'} failed ({code}): {msg}")
