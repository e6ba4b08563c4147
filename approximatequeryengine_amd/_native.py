"""ctypes binding of include/aqe_hip.h (libaqe_hip.so).  No CPU fallback: if the library is missing it
is built with hipcc; if that is impossible, importing the compute path raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

from .build import LIB, build_native

# ---- enums of include/aqe_hip.h ------------------------------------------------------------------
OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_NO_TABLE, ERR_IO, ERR_CAPACITY, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7

M_EXACT, M_MEMORY_STRIDE, M_ADDRESS_ARITHMETIC, M_RANDOM_POINTER, M_BLOCK, M_PAGE, M_PARALLEL_BLOCK = range(7)
M_OPTIMIZED_CLT, M_CLT_DUAL_POINTER, M_FAST_POINTER, M_SLOW_POINTER, M_DUAL_POINTER = 7, 8, 9, 10, 11
M_PARALLEL_POINTER, M_REGION_STRIDE, M_RANDOM_START_STRIDE, M_ADAPTIVE_BLOCK, M_STRATIFIED_BLOCK = 12, 13, 14, 15, 16
M_ROWID_MOD = 17
M_RANDOM_DEVICE = 18
M_DIRECT_ACCESS = 19
M_OPTIMIZED_SEQUENTIAL = 20
GROUP_REGION, GROUP_PRODUCT = 1, 2

SUM, AVG, COUNT = 0, 1, 2
EST_CLI, EST_CPP, EST_RAW = 0, 1, 2
Q_NO_TOPUP = 1
Q_NO_PERSIST = 2
Q_FORCE_PERSIST = 4
Q_NO_LAYOUT = 8
Q_SHARE_GPU = 16
Q_NO_LEAN = 32
Q_FORCE_LEAN = 64
KERNEL_ROUND, KERNEL_SWEEP_PERSIST, KERNEL_SWEEP_LEAN, KERNEL_SWEEP_MULTI, KERNEL_SWEEP_LEAN_MULTI = 0, 1, 2, 3, 4
KERNEL_NAMES = {0: "k_round", 1: "k_sweep_persist", 2: "k_sweep_lean", 3: "k_sweep_multi", 4: "k_sweep_lean_multi", 5: "k_indexed", 6: "k_permuted"}
F_TOPUP = 1
F_PAIR = 2
STAGE_KEEP_AOS = 1
MOMENT_VEC = 8
COMM_ID_BYTES = 128
MAILBOX_HANDLE_BYTES = 64
MAILBOX_MAX_DOUBLES = 4096


class Query(C.Structure):
    _fields_ = [
        ("method", C.c_int32), ("agg", C.c_int32), ("convention", C.c_int32), ("num_threads", C.c_int32),
        ("sample_percent", C.c_double), ("stride_bytes", C.c_uint64), ("block_size", C.c_uint64),
        ("seed", C.c_uint64), ("step_size", C.c_int32), ("check_interval", C.c_int32),
        ("confidence_level", C.c_double), ("max_error_percent", C.c_double), ("has_where", C.c_int32),
        ("reserved0", C.c_int32), ("where_min", C.c_double), ("where_max", C.c_double),
        ("clt_round0", C.c_uint64), ("clt_growth", C.c_uint32), ("flags", C.c_uint32),
        ("visible_rows", C.c_uint64), ("block_size_max", C.c_uint64), ("row_lo", C.c_uint64), ("row_hi", C.c_uint64),
    ]


class Result(C.Structure):
    _fields_ = [
        ("value", C.c_double), ("ci_lower", C.c_double), ("ci_upper", C.c_double), ("margin", C.c_double),
        ("sum", C.c_double), ("sumsq", C.c_double), ("mean", C.c_double), ("m2", C.c_double),
        ("n", C.c_uint64), ("visited", C.c_uint64), ("topup", C.c_uint64), ("converged", C.c_int32),
        ("rounds", C.c_int32), ("kernel_ms", C.c_double), ("bytes_algorithmic", C.c_uint64),
        ("device_status", C.c_int32), ("topup_pending", C.c_int32),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class Family(C.Structure):
    _fields_ = [("row0", C.c_uint64), ("pitch", C.c_uint64), ("seg_len", C.c_uint64), ("step", C.c_uint64),
                ("ord_lo", C.c_uint64), ("ord_hi", C.c_uint64), ("row0_b", C.c_uint64), ("ord_lo_b", C.c_uint64),
                ("ord_hi_b", C.c_uint64), ("group", C.c_uint32), ("flags", C.c_uint32)]


class GroupResult(C.Structure):
    _fields_ = [("key", C.c_int64), ("n", C.c_uint64), ("visited", C.c_uint64), ("sum", C.c_double), ("sumsq", C.c_double),
                ("mean", C.c_double), ("value", C.c_double), ("ci_lower", C.c_double), ("ci_upper", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class TableInfo(C.Structure):
    _fields_ = [("global_rows", C.c_uint64), ("shard_lo", C.c_uint64), ("local_rows", C.c_uint64),
                ("shift", C.c_double), ("has_aos", C.c_int32), ("device_id", C.c_int32), ("hbm_bytes", C.c_uint64),
                ("view_bytes", C.c_uint64), ("n_views", C.c_uint32), ("view_evictions", C.c_uint32), ("view_fallbacks", C.c_uint32),
                ("reserved", C.c_uint32)]


class StageStats(C.Structure):
    _fields_ = [("total_ms", C.c_double), ("device_alloc_ms", C.c_double), ("pinned_alloc_ms", C.c_double), ("fill_ms", C.c_double),
                ("wait_ms", C.c_double), ("host_bytes", C.c_uint64), ("link_bytes", C.c_uint64), ("chunks", C.c_uint32), ("fill_threads", C.c_uint32)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class AqeError(RuntimeError):
    """A C-ABI call failed; mirrors pybind11 turning C++ exceptions into RuntimeError."""

    def __init__(self, status: int, message: str):
        super().__init__(f"[{status}] {message}")
        self.status = status


_lib = None


def lib() -> C.CDLL:
    """Load (building first if needed) libaqe_hip.so.  Raises if it cannot be produced."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own HIP runtime.  When both live in one process torch's copy must be the
    # one that gets loaded (loading /opt/rocm's first leaves torch without a device), so import torch
    # before dlopen-ing the library.  torch stays optional: without it the system runtime is used.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = Path(os.environ.get("AQE_HIP_LIB", LIB))
    if not path.exists() or "AQE_HIP_LIB" not in os.environ:
        path = build_native()
    L = C.CDLL(str(path))
    vp, u64, u32, i32, dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32, C.c_double
    P = C.POINTER
    sig = {
        "aqe_abi_version": (C.c_int, []),
        "aqe_create": (C.c_int, [C.c_int, P(vp)]),
        "aqe_destroy": (None, [vp]),
        "aqe_last_error": (C.c_char_p, [vp]),
        "aqe_status_string": (C.c_char_p, [C.c_int]),
        "aqe_stage_records": (C.c_int, [vp, vp, u64, u64, u64, u32]),
        "aqe_stage_file": (C.c_int, [vp, C.c_char_p, u64, u64, u32]),
        "aqe_file_rows": (C.c_int, [C.c_char_p, P(u64)]),
        "aqe_last_stage_stats": (C.c_int, [vp, P(StageStats)]),
        "aqe_save_file": (C.c_int, [vp, C.c_char_p]),
        "aqe_generate_synthetic": (C.c_int, [vp, u64, u64, u64, u64, u32]),
        "aqe_attach_device": (C.c_int, [vp, vp, vp, u64, u64, u64, dbl]),
        "aqe_set_shift": (C.c_int, [vp, dbl]),
        "aqe_table_info_get": (C.c_int, [vp, P(TableInfo)]),
        "aqe_key_range_rows": (C.c_int, [vp, C.c_int64, C.c_int64, P(u64), P(u64)]),
        "aqe_key_range_counts": (C.c_int, [vp, C.c_int64, C.c_int64, P(u64), P(u64)]),
        "aqe_release_table": (C.c_int, [vp]),
        "aqe_device_malloc": (C.c_int, [vp, C.c_size_t, P(vp)]),
        "aqe_device_free": (C.c_int, [vp, vp]),
        "aqe_device_read": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
        "aqe_device_write": (C.c_int, [vp, vp, vp, C.c_size_t, vp]),
        "aqe_query_defaults": (None, [P(Query)]),
        "aqe_plan_families": (C.c_int, [P(Query), u64, u64, u64, u32, P(Family), u32, P(u32), P(u32), P(u64)]),
        "aqe_plan_random_indices": (C.c_int, [u64, dbl, u32, u64, u64, P(u64), u64, P(u64)]),
        "aqe_plan_row_list": (C.c_int, [P(Query), u64, u64, u64, P(u64), u64, P(u64)]),
        "aqe_plan_adaptive_families": (C.c_int, [P(Query), u64, P(dbl), P(Family), u32, P(u32), P(u64)]),
        "aqe_parse_where": (C.c_int, [C.c_char_p, P(dbl), P(dbl)]),
        "aqe_confidence_heuristic": (dbl, [dbl, u64]),
        "aqe_error_to_sample_percent": (dbl, [dbl]),
        "aqe_reduce": (C.c_int, [vp, P(Query), P(Result)]),
        "aqe_reduce_grouped": (C.c_int, [vp, P(Query), C.c_int, P(GroupResult), u32, P(u32)]),
        "aqe_group_key_range": (C.c_int, [vp, C.c_int, P(C.c_int32), P(C.c_int32)]),
        "aqe_grouped_enqueue_bins": (C.c_int, [vp, P(Query), C.c_int, C.c_int32, u32, vp, vp]),
        "aqe_grouped_finish": (C.c_int, [vp, P(Query), C.c_int32, u32, vp, vp, P(GroupResult), u32, P(u32)]),
        "aqe_gather": (C.c_int, [vp, P(Query), vp, u64, P(u64)]),
        "aqe_mailbox_create": (C.c_int, [vp, C.c_int, C.c_int, P(vp)]),
        "aqe_mailbox_handle": (C.c_int, [vp, vp]),
        "aqe_mailbox_connect": (C.c_int, [vp, vp]),
        "aqe_mailbox_connect_local": (C.c_int, [P(vp), C.c_int]),
        "aqe_mailbox_all_reduce_sum": (C.c_int, [vp, vp, u64, vp]),
        "aqe_mailbox_status": (C.c_int, [vp, P(u32)]),
        "aqe_mailbox_info": (C.c_int, [vp, P(C.c_int), P(C.c_int)]),
        "aqe_comm_create_mailbox": (C.c_int, [vp, vp, P(vp)]),
        "aqe_mailbox_destroy": (None, [vp]),
        "aqe_plan_create": (C.c_int, [vp, P(Query), P(vp)]),
        "aqe_plan_create_families": (C.c_int, [vp, P(Query), P(Family), u32, u64, C.c_int, P(vp)]),
        "aqe_zone_moments": (C.c_int, [vp, P(dbl)]),
        "aqe_set_zone_variances": (C.c_int, [vp, P(dbl)]),
        "aqe_sorted_counts": (C.c_int, [vp, P(dbl), u32, P(u64), P(u64)]),
        "aqe_plan_destroy": (None, [vp]),
        "aqe_plan_rounds": (C.c_int, [vp, P(u32), P(i32)]),
        "aqe_plan_enqueue_round": (C.c_int, [vp, u32, vp, vp]),
        "aqe_plan_enqueue_update": (C.c_int, [vp, u32, vp, vp]),
        "aqe_plan_enqueue_finalize": (C.c_int, [vp, vp]),
        "aqe_plan_enqueue_all": (C.c_int, [vp, vp]),
        "aqe_plan_totals_len": (C.c_int, [vp, P(u32)]),
        "aqe_plan_enqueue_sweep_totals": (C.c_int, [vp, vp, vp]),
        "aqe_plan_enqueue_replay": (C.c_int, [vp, vp, vp]),
        "aqe_batch_create": (C.c_int, [P(vp), u32, P(vp)]),
        "aqe_batch_destroy": (None, [vp]),
        "aqe_batch_enqueue_sweeps": (C.c_int, [vp, vp, u64]),
        "aqe_batch_join": (C.c_int, [vp, vp]),
        "aqe_batch_enqueue_replays": (C.c_int, [vp, vp, u64, vp]),
        "aqe_batch_fetch": (C.c_int, [vp, P(Result)]),
        "aqe_batch_enqueue_all": (C.c_int, [vp, vp]),
        "aqe_batch_set_profiling": (C.c_int, [vp, C.c_int]),
        "aqe_batch_launch_info": (C.c_int, [vp, P(C.c_float), P(u64), P(u32)]),
        "aqe_comm_unique_id": (C.c_int, [vp]),
        "aqe_comm_create": (C.c_int, [vp, vp, C.c_int, C.c_int, P(vp)]),
        "aqe_comm_create_all": (C.c_int, [P(vp), C.c_int, P(vp)]),
        "aqe_comm_destroy": (None, [vp]),
        "aqe_comm_info": (C.c_int, [vp, P(C.c_int), P(C.c_int)]),
        "aqe_comm_all_reduce_sum": (C.c_int, [vp, vp, u64, vp]),
        "aqe_comm_all_reduce_max": (C.c_int, [vp, vp, u64, vp]),
        "aqe_comm_group_start": (C.c_int, []),
        "aqe_comm_group_end": (C.c_int, []),
        "aqe_plan_run_sharded": (C.c_int, [vp, vp, vp, vp, P(Result)]),
        "aqe_batch_run_sharded": (C.c_int, [vp, vp, vp, u64, u32, vp]),
        "aqe_plan_reset": (C.c_int, [vp, vp]),
        "aqe_plan_fetch": (C.c_int, [vp, P(Result), vp]),
        "aqe_plan_last_kernel_ms": (C.c_int, [vp, P(C.c_float)]),
        "aqe_plan_set_profiling": (C.c_int, [vp, C.c_int]),
        "aqe_plan_launch_ms": (C.c_int, [vp, P(C.c_float), u32, P(u32)]),
        "aqe_plan_launch_samples": (C.c_int, [vp, P(u64), u32, P(u32)]),
        "aqe_plan_last_kernel": (C.c_int, [vp, P(C.c_int)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    if L.aqe_abi_version() != 2:
        raise ImportError("libaqe_hip.so ABI version mismatch")
    _lib = L
    return L


def default_query(**kw) -> Query:
    q = Query()
    lib().aqe_query_defaults(C.byref(q))
    for k, v in kw.items():
        if not hasattr(q, k):
            raise TypeError(f"aqe_query has no field {k!r}")
        setattr(q, k, v)
    return q


def check(status: int, ctx=None):
    if status != OK:
        msg = lib().aqe_last_error(ctx)
        text = msg.decode() if msg else ""
        raise AqeError(status, text or lib().aqe_status_string(status).decode())


def plan_families(q: Query, n_global: int, lo: int = 0, hi: int | None = None, round: int = 0):
    """Host-side plan of `q` (no GPU): (families, rounds, global_samples)."""
    L = lib()
    hi = n_global if hi is None else hi
    n, rounds, samples = C.c_uint32(), C.c_uint32(), C.c_uint64()
    check(L.aqe_plan_families(C.byref(q), n_global, lo, hi, round, None, 0, C.byref(n), C.byref(rounds), C.byref(samples)))
    fams = (Family * max(n.value, 1))()
    check(L.aqe_plan_families(C.byref(q), n_global, lo, hi, round, fams, n.value, C.byref(n), C.byref(rounds), C.byref(samples)))
    return list(fams[: n.value]), rounds.value, samples.value


def plan_adaptive_families(q: Query, n_global: int, zone_var):
    """Host-side plan of adaptive_block_sample from the ten zone variances: (families, global_samples)."""
    L = lib()
    zv = (C.c_double * 10)(*[float(v) for v in zone_var])
    n, samples = C.c_uint32(), C.c_uint64()
    check(L.aqe_plan_adaptive_families(C.byref(q), n_global, zv, None, 0, C.byref(n), C.byref(samples)))
    fams = (Family * max(n.value, 1))()
    check(L.aqe_plan_adaptive_families(C.byref(q), n_global, zv, fams, n.value, C.byref(n), C.byref(samples)))
    return list(fams[: n.value]), samples.value


def plan_random_indices(n_global: int, pct: float, seed: int, lo: int = 0, hi: int | None = None):
    import numpy as np
    L = lib()
    hi = n_global if hi is None else hi
    n = C.c_uint64()
    check(L.aqe_plan_random_indices(n_global, pct, seed, lo, hi, None, 0, C.byref(n)))
    out = np.zeros(max(n.value, 1), dtype=np.uint64)
    check(L.aqe_plan_random_indices(n_global, pct, seed, lo, hi, out.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)))
    return out[: n.value]


def plan_row_list(q: Query, n_global: int, lo: int = 0, hi: int | None = None):
    """Host-side row list (no GPU) of a sampler without families: RANDOM_POINTER, DIRECT_ACCESS, OPTIMIZED_SEQUENTIAL."""
    import numpy as np
    L = lib()
    hi = n_global if hi is None else hi
    n = C.c_uint64()
    check(L.aqe_plan_row_list(C.byref(q), n_global, lo, hi, None, 0, C.byref(n)))
    out = np.zeros(max(n.value, 1), dtype=np.uint64)
    check(L.aqe_plan_row_list(C.byref(q), n_global, lo, hi, out.ctypes.data_as(C.POINTER(C.c_uint64)), n.value, C.byref(n)))
    return out[: n.value]
