"""ctypes binding of libsimrank_hip.so (the C ABI declared in include/simrank_hip.h).

There is no CPU fallback: if the shared library is missing, or no gfx950 device is
visible when a device call is made, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SIMRANK_LIB: an experiment build of the same ABI, tools/build_variant.sh; never set in production)
LIB_PATH = os.environ.get("SIMRANK_LIB") or os.path.join(_HERE, "libsimrank_hip.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "simrank_hip.h")


class SimRankHipError(RuntimeError):
    """A call into libsimrank_hip.so failed."""


class Epilogue(C.Structure):
    """struct simrank_epilogue (include/simrank_hip.h)."""
    _fields_ = [
        ("coef", C.c_float),
        ("lbd", C.c_float),
        ("evidence", C.c_void_p),
        ("ld_evidence", C.c_int64),
        ("apriori", C.c_void_p),
        ("ld_apriori", C.c_int64),
        ("previous", C.c_void_p),
        ("ld_previous", C.c_int64),
        ("eps", C.c_double),
        ("n_changed", C.c_void_p),
        ("diag_col0", C.c_int64),
        ("set_diag", C.c_int32),
        ("symmetric", C.c_int32),
        ("restrict_support", C.c_int32),
        ("count_any", C.c_int32),
    ]


ABI_VERSION = 7          # SIMRANK_ABI_VERSION of include/simrank_hip.h
# the console hooks of the C-level loops (simrank_progress_fn)
PROGRESS_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_int32)

_vp, _i64, _i32, _int = C.c_void_p, C.c_int64, C.c_int32, C.c_int
_pvp = C.POINTER(C.c_void_p)

# name -> argtypes (restype is int unless listed in _RESTYPES)
PROTOTYPES = {
    "simrank_abi_version": [],
    "simrank_last_error": [],
    "simrank_device_count": [C.POINTER(_int)],
    "simrank_set_device": [_int],
    "simrank_device_info": [_int, C.c_char_p, _int, C.POINTER(_i64), C.POINTER(_int),
                            C.c_char_p, _int],
    "simrank_malloc": [_pvp, C.c_size_t],
    "simrank_free": [_vp],
    "simrank_pool_trim": [_int],
    "simrank_pool_stats": [_int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "simrank_memset": [_vp, _int, C.c_size_t, _vp],
    "simrank_memcpy_h2d": [_vp, _vp, C.c_size_t, _vp],
    "simrank_memcpy_d2h": [_vp, _vp, C.c_size_t, _vp],
    "simrank_memcpy_d2d": [_vp, _vp, C.c_size_t, _vp],
    "simrank_download_f64": [_vp, _i64, _vp, _i64, _i64, _i64, _vp],
    "simrank_read_counters": [_vp, _i32, C.POINTER(C.c_ulonglong), _vp],
    "simrank_handback_f64": [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _i32, _vp],
    "simrank_counters_create": [_pvp],
    "simrank_counters_destroy": [_vp],
    "simrank_counters_fetch": [_vp, _vp, _i32, _i32, _vp],
    "simrank_counters_wait": [_vp, _i32, C.POINTER(C.c_ulonglong)],
    "simrank_stream_create": [_pvp],
    "simrank_stream_destroy": [_vp],
    "simrank_stream_synchronize": [_vp],
    "simrank_event_create": [_pvp],
    "simrank_event_destroy": [_vp],
    "simrank_event_record": [_vp, _vp],
    "simrank_event_synchronize": [_vp],
    "simrank_event_elapsed_ms": [_vp, _vp, C.POINTER(C.c_float)],
    "simrank_graph_create": [_i64, _i64, _i64, _vp, _vp, _vp, _pvp],
    "simrank_graph_create_counting": [_i64, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _vp, _i64, _i64, _vp, C.POINTER(_vp)],
    "simrank_graph_destroy": [_vp],
    "simrank_graph_shape": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "simrank_fill_identity": [_vp, _i64, _i64, _i64, _i64, _vp],
    "simrank_spmm": [_vp, _vp, _i64, _i64, _vp, _i64, _i32, _i64, _i64, C.POINTER(Epilogue), _vp],
    "simrank_epilogue_apply": [_vp, _i64, _vp, _i64, _i64, _i64, C.POINTER(Epilogue), _vp],
    "simrank_topk_rows": [_vp, _i64, _i64, _i64, _i64, _i32, _i32, _vp, _vp, _vp],
    "simrank_topk_rows_ids": [_vp, _i64, _i64, _i64, _i64, _vp, _i32, _i32, _vp, _vp, _vp],
    "simrank_permute": [_vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _i32, _vp],
    "simrank_evidence_live_segments": [_vp, _i64, _i64, _i64, _i64, C.POINTER(_i64), C.POINTER(_i64), _vp],
    "simrank_spmm_shard": [_vp, _vp, _i64, _vp, _i64, C.POINTER(Epilogue), _i32, _i32, _vp, _i64, _vp],
    "simrank_shard_unpack": [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _vp],
    "simrank_spmm_shard_stage": [_vp, _vp, _i64, _vp, _i64, C.POINTER(Epilogue), _i32, _i32, _vp, _i64, _i32, _i32,
                                 _i32, _vp],
    "simrank_shard_unpack_stage": [_vp, _i64, _vp, _i64, _i32, _i32, _i64, _i32, _i32, _vp],
    "simrank_fill_identity_blocked": [_vp, _i64, _i64, _i64, _i64, _vp],
    "simrank_spmm_blocked": [_vp, _vp, _i64, _i64, _vp, _i64, _i32, C.POINTER(Epilogue), _vp],
    "simrank_plan_topk": [_vp, _i32, _i32, _vp, _vp],
    "simrank_plan_rows_f32": [_vp, _vp, _i32, _vp, _i64],
    "simrank_plan_run_cb": [_vp, C.c_int32, C.c_double, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "simrank_plan_evidence_u8": [_vp, _vp, _i64],
    "simrank_plan_set_timing": [_vp, C.c_int32],
    "simrank_plan_leg_times": [_vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)],
    "simrank_plan_trim": [_vp],
    "simrank_biplan_run_cb": [_vp, C.c_int32, C.c_double, _vp, _vp, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "simrank_biplan_topk": [_vp, _i32, _i32, _i32, _vp, _vp],
    "simrank_biplan_rows_f32": [_vp, _i32, _vp, _i32, _vp, _i64],
    "simrank_biplan_evidence_u8": [_vp, _i32, _vp, _i64],
    "simrank_biplan_trim": [_vp],
    "simrank_biplan_create": [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "simrank_biplan_reset": [_vp],
    "simrank_biplan_step": [_vp, C.c_double, _i32, _vp, _vp],
    "simrank_biplan_run": [_vp, _i32, C.c_double, _vp, _vp],
    "simrank_biplan_result_f64": [_vp, _i32, _vp, _i64],
    "simrank_biplan_destroy": [_vp],
    "simrank_fill_identity_blocked_h16": [_vp, _i64, _i64, _i64, _i64, C.c_float, _vp],
    "simrank_spmm_blocked_h16": [_vp, _vp, _i64, _i64, _vp, _i64, _i32, C.POINTER(Epilogue), _i64, C.c_float, _vp],
    "simrank_widen_blocked_h16": [_vp, _i64, _vp, _i64, _i64, _i64, C.c_float, _vp],
    "simrank_narrow_h16": [_vp, _vp, _i64, C.c_float, _vp],
    "simrank_widen_h16": [_vp, _vp, _i64, C.c_float, _vp],
    "simrank_epilogue_apply_blocked": [_vp, _vp, _i64, _i64, _i64, C.POINTER(Epilogue), _vp],
    "simrank_topk_rows_blocked": [_vp, _i64, _i64, _i64, _i64, _vp, _i32, _i32, _vp, _vp, _vp],
    "simrank_evidence_counts_blocked": [_vp, _i64, _i64, _vp, _i64, _vp],
    "simrank_permute_layout": [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _i64, _vp, _vp, _i32, _vp],
    "simrank_evidence_counts": [_vp, _i64, _i64, _vp, _i64, _vp],
    "simrank_graph_densify": [_vp, _vp, _i64, _vp],
    "simrank_gemm_nt": [_i64, _i64, _i64, _vp, _i64, _vp, _i64, _vp, _i64,
                        C.POINTER(Epilogue), _vp],
    "simrank_graph_dense_stats": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "simrank_graph_set_dense_terms": [_vp, C.c_int32],
    "simrank_graph_fused_stats": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64)],
    "simrank_dense_part": [_vp, _vp, _i64, _i64, _vp],
    "simrank_plan_create": [_i64, _i64, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)],
    "simrank_plan_reset": [_vp],
    "simrank_plan_step": [_vp, C.c_double, C.c_int32, C.POINTER(_i64)],
    "simrank_plan_run": [_vp, C.c_int32, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "simrank_plan_result": [_vp, _vp, _i64],
    "simrank_plan_result_f64": [_vp, _vp, _i64],
    "simrank_plan_info": [_vp, C.POINTER(_i64), C.POINTER(C.c_int32), C.POINTER(_vp)],
    "simrank_plan_destroy": [_vp],
    "simrank_comm_unique_id": [_vp],
    "simrank_comm_create": [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)],
    "simrank_comm_adopt": [_vp, C.c_int32, C.c_int32, C.POINTER(_vp)],
    "simrank_comm_local_group": [C.c_int32, C.POINTER(_vp)],
    "simrank_comm_thread_group": [C.c_int32, C.POINTER(_vp)],
    "simrank_comm_destroy": [_vp],
    "simrank_shardplan_create": [_i64, _i64, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)],
    "simrank_shardplan_reset": [C.POINTER(_vp), C.c_int32],
    "simrank_shardplan_step": [C.POINTER(_vp), C.c_int32, C.c_double, C.c_int32, C.POINTER(_i64)],
    "simrank_shardplan_run": [C.POINTER(_vp), C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "simrank_shardplan_block_f64": [_vp, _vp, _i64],
    "simrank_shardplan_columns": [_vp, _vp],
    "simrank_shardplan_result_f64": [C.POINTER(_vp), C.c_int32, C.c_int32, _vp, _i64],
    "simrank_shardplan_topk": [C.POINTER(_vp), C.c_int32, C.c_int32, C.c_int32, C.c_int32, _vp, _vp],
    "simrank_shardplan_info": [_vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(C.c_int32),
                               C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "simrank_shardplan_destroy": [_vp],
    "simrank_shardplan_set_timing": [_vp, C.c_int32],
    "simrank_shardplan_timings": [_vp, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int32)],
    "simrank_shardbiplan_create": [_i64, _i64, _i64, _vp, _vp, _vp, _vp, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp,
                                   C.POINTER(_vp)],
    "simrank_shardbiplan_side": [_vp, C.c_int32, C.POINTER(_vp)],
    "simrank_shardbiplan_reset": [C.POINTER(_vp), C.c_int32],
    "simrank_shardbiplan_step": [C.POINTER(_vp), C.c_int32, C.c_double, C.c_int32, C.POINTER(_i64), C.POINTER(_i64)],
    "simrank_shardbiplan_run": [C.POINTER(_vp), C.c_int32, C.c_int32, C.c_double, C.POINTER(C.c_int32), C.POINTER(C.c_int32)],
    "simrank_shardbiplan_destroy": [_vp],
    "simrank_set_tuning": [C.c_char_p, _i64],
    "simrank_get_tuning": [C.c_char_p, C.POINTER(_i64)],
}
_RESTYPES = {"simrank_last_error": C.c_char_p}

_lib = None


def load() -> C.CDLL:
    """Load the library once; raise ImportError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP engine is not built.  Build it with "
            "`make -C simrank_amd/csrc` (or `python -c 'import __graft_entry__ as g; "
            "g.build()'`).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError = symbol missing from the .so
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    if lib.simrank_abi_version() != ABI_VERSION:
        raise ImportError("libsimrank_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().simrank_last_error()
        raise SimRankHipError(f"{what or 'simrank call'} failed ({rc}): "
                              f"{msg.decode() if msg else '?'}")


def device_count() -> int:
    """Number of visible HIP devices (0 when there is none; never raises)."""
    n = C.c_int(0)
    load().simrank_device_count(C.byref(n))
    return n.value
