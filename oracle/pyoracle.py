"""oracle/pyoracle.py — TEST INFRASTRUCTURE ONLY.

ctypes front-end to the two checkers built by oracle/Makefile:

* ``Oracle``  -> oracle/_build/libaqe_oracle.so, our plain-C restatement (oracle/aqe_oracle.c);
* ``Ref``     -> oracle/_ref/libaqe_ref.so, the reference's own C++ hot path compiled from
  /root/reference by oracle/ref_harness.cpp (exists only where that tree was present at build time).

Allowed importers: tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg, oracle/make_golden.py.
The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
ORACLE_SO = HERE / "_build" / "libaqe_oracle.so"
REF_SO = HERE / "_ref" / "libaqe_ref.so"

#: 32-byte row of DB.hpp:17-27
REC = np.dtype(
    [("id", "<i8"), ("amount", "<f8"), ("region", "<i4"), ("product_id", "<i4"), ("timestamp", "<i8")]
)
assert REC.itemsize == 32

SUM, AVG, COUNT = 0, 1, 2
MAX_WORKERS = 256


def build(ref: bool = True, quiet: bool = True) -> None:
    """Compile the C restatement and, when /root/reference is present, the reference harness."""
    out = subprocess.DEVNULL if quiet else None
    subprocess.check_call(["make", "-C", str(HERE), "all"], stdout=out)
    if ref:
        subprocess.check_call(["make", "-C", str(HERE), "ref"], stdout=out)


class Moments(C.Structure):
    _fields_ = [("n", C.c_uint64), ("sum", C.c_double), ("sumsq", C.c_double), ("mean", C.c_double),
                ("m2", C.c_double)]

    def as_dict(self):
        return {"n": int(self.n), "sum": self.sum, "sumsq": self.sumsq, "mean": self.mean, "m2": self.m2}


class CltWorker(C.Structure):
    _fields_ = [("first", C.c_uint64), ("end", C.c_uint64), ("step", C.c_uint64), ("count", C.c_uint64),
                ("is_fast", C.c_int), ("group", C.c_int)]


class CltPlan(C.Structure):
    _fields_ = [("base", C.c_int), ("n_workers", C.c_int), ("n_fast", C.c_int), ("z", C.c_double),
                ("w", CltWorker * MAX_WORKERS)]


class CltResult(C.Structure):
    _fields_ = [("all", Moments), ("fast", Moments), ("slow", Moments), ("final", Moments),
                ("converged", C.c_int), ("rounds", C.c_int), ("topup", C.c_uint64)]


_u64p = C.POINTER(C.c_uint64)
_i64p = C.POINTER(C.c_int64)
_f64p = C.POINTER(C.c_double)


def _ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty)


class Oracle:
    """Thin, typed wrapper over libaqe_oracle.so."""

    def __init__(self, path: os.PathLike | None = None):
        path = Path(path) if path else ORACLE_SO
        if not path.exists():
            build(ref=False)
        L = self.lib = C.CDLL(str(path))
        L.aqo_splitmix64_at.restype = C.c_uint64
        L.aqo_splitmix64_at.argtypes = [C.c_uint64, C.c_uint64]
        L.aqo_synth_amount.restype = C.c_double
        L.aqo_synth_amount.argtypes = [C.c_uint64, C.c_uint64]
        L.aqo_synth_fill.restype = None
        L.aqo_synth_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint64]
        for name, extra in {
            "aqo_idx_memory_stride": [C.c_uint64],
            "aqo_idx_address_arithmetic": [],
            "aqo_idx_random_pointer": [C.c_uint32],
            "aqo_idx_direct_access": [],
            "aqo_idx_optimized_sequential": [C.c_uint32],
            "aqo_idx_block": [C.c_uint64],
            "aqo_idx_page": [C.c_uint64],
            "aqo_idx_parallel_block": [C.c_uint64, C.c_int],
            "aqo_idx_optimized_clt": [C.c_int],
            "aqo_idx_fast_pointer": [C.c_int],
            "aqo_idx_dual_pointer": [],
            "aqo_idx_parallel_pointer": [C.c_int],
        }.items():
            f = getattr(L, name)
            f.restype = C.c_int64
            f.argtypes = [C.c_uint64, C.c_double] + extra + [_u64p, C.c_int64]
        L.aqo_idx_adaptive_block.restype = C.c_int64
        L.aqo_idx_adaptive_block.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_uint64, C.c_uint64, _f64p, _u64p, C.c_int64]
        L.aqo_idx_stratified_block.restype = C.c_int64
        L.aqo_idx_stratified_block.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_uint64, C.c_int, _u64p, C.c_int64]
        L.aqo_idx_random_start_stride.restype = C.c_int64
        L.aqo_idx_random_start_stride.argtypes = [C.c_uint64, C.c_double, C.c_uint64, C.c_uint64, _u64p, _u64p, C.c_int64]
        L.aqo_idx_region_stride.restype = C.c_int64
        L.aqo_idx_region_stride.argtypes = [C.c_uint64, C.c_double, C.c_int, C.c_uint64, _u64p, C.c_int,
                                            _u64p, C.c_int64]
        L.aqo_moments_idx.restype = None
        L.aqo_moments_idx.argtypes = [C.c_void_p, _u64p, C.c_int64, C.c_int, C.c_double, C.c_double,
                                      C.POINTER(Moments)]
        L.aqo_moments_range.restype = None
        L.aqo_moments_range.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_double, C.c_double,
                                        C.POINTER(Moments)]
        L.aqo_estimate_cli.restype = C.c_double
        L.aqo_estimate_cli.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double]
        L.aqo_estimate_cpp.restype = C.c_double
        L.aqo_estimate_cpp.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_uint64, C.c_double]
        L.aqo_ci_cli.restype = C.c_double
        L.aqo_ci_cli.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_double, C.c_double, _f64p, _f64p]
        L.aqo_margin_moments.restype = C.c_double
        L.aqo_margin_moments.argtypes = [C.c_uint64, C.c_double, C.c_double]
        L.aqo_confidence_heuristic.restype = C.c_double
        L.aqo_confidence_heuristic.argtypes = [C.c_double, C.c_uint64]
        L.aqo_error_to_percent.restype = C.c_double
        L.aqo_error_to_percent.argtypes = [C.c_double]
        L.aqo_clt_zscore.restype = C.c_double
        L.aqo_clt_zscore.argtypes = [C.c_double]
        L.aqo_clt_error_percent.restype = C.c_double
        L.aqo_clt_error_percent.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_double]
        L.aqo_clt_fast_rule.restype = C.c_int
        L.aqo_clt_fast_rule.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double]
        L.aqo_clt_slow_rule.restype = C.c_int
        L.aqo_clt_slow_rule.argtypes = [C.c_uint64, C.c_double, C.c_uint64, C.c_double, C.c_double, C.c_int]
        L.aqo_clt_make_plan.restype = C.c_int
        L.aqo_clt_make_plan.argtypes = [C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int,
                                        C.POINTER(CltPlan)]
        L.aqo_clt_run.restype = C.c_int
        L.aqo_clt_run.argtypes = [C.c_void_p, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int,
                                  C.c_double, C.c_uint64, C.c_uint32, C.POINTER(CltResult), _u64p,
                                  C.c_int64, _i64p]
        L.aqo_clt_round_partial.restype = None
        L.aqo_clt_round_partial.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.POINTER(CltPlan),
                                            C.c_uint64, C.c_uint64, C.POINTER(Moments), C.POINTER(Moments)]
        L.aqo_file_write.restype = C.c_int
        L.aqo_file_write.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64]
        L.aqo_file_count.restype = C.c_int64
        L.aqo_file_count.argtypes = [C.c_char_p]
        L.aqo_file_read.restype = C.c_int64
        L.aqo_file_read.argtypes = [C.c_char_p, C.c_void_p, C.c_uint64, C.c_uint64]
        i64p = C.POINTER(C.c_int64)
        L.aqo_group_rowid_mod.restype = C.c_int64
        L.aqo_group_rowid_mod.argtypes = [C.c_void_p, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, i64p, _u64p,
                                          C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64]
        L.aqo_group_idx.restype = C.c_int64
        L.aqo_group_idx.argtypes = [C.c_void_p, _u64p, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double, i64p, _u64p,
                                    C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_int64]
        L.aqo_group_ci.restype = None
        L.aqo_group_ci.argtypes = [C.c_int, C.c_uint64, C.c_double, C.c_double, C.c_int, C.c_int] + [C.POINTER(C.c_double)] * 3
        L.aqo_mt19937_stream.restype = None
        L.aqo_mt19937_stream.argtypes = [C.c_uint32, C.POINTER(C.c_uint32), C.c_int]

    # ---- data ----
    def synth(self, n: int, seed: int = 42, first: int = 0) -> np.ndarray:
        rows = np.zeros(n, dtype=REC)
        if n:
            self.lib.aqo_synth_fill(rows.ctypes.data, first, n, seed)
        return rows

    # ---- GROUP BY (EXE:202-321) ----
    def group(self, rows, group_col, sample_percent=None, idx=None, where=None, cap=4096):
        """Per-group (key, n, sum, sumsq), keys ascending: over the rowid %% step sample (sample_percent) or an index list."""
        keys = np.zeros(cap, dtype=np.int64); n = np.zeros(cap, dtype=np.uint64)
        s = np.zeros(cap, dtype=np.float64); q = np.zeros(cap, dtype=np.float64)
        dp = C.POINTER(C.c_double)
        hw, lo, hi = (1, float(where[0]), float(where[1])) if where else (0, 0.0, 0.0)
        args = (group_col, hw, lo, hi, keys.ctypes.data_as(C.POINTER(C.c_int64)), _ptr(n, _u64p), s.ctypes.data_as(dp), q.ctypes.data_as(dp), cap)
        if idx is not None:
            idx = np.ascontiguousarray(idx, dtype=np.uint64)
            g = self.lib.aqo_group_idx(rows.ctypes.data, _ptr(idx, _u64p), len(idx), *args)
        else:
            g = self.lib.aqo_group_rowid_mod(rows.ctypes.data, len(rows), int(sample_percent), *args)
        assert g <= cap
        return [(int(keys[i]), int(n[i]), float(s[i]), float(q[i])) for i in range(g)]

    def group_ci(self, agg, count, total, sumsq, sample_percent, reference_sum=False):
        v, lo, hi = C.c_double(), C.c_double(), C.c_double()
        self.lib.aqo_group_ci(agg, count, total, sumsq, int(sample_percent), int(reference_sum), C.byref(v), C.byref(lo), C.byref(hi))
        return v.value, lo.value, hi.value

    # ---- index sets ----
    def _idx(self, fname: str, n_rows: int, pct: float, *extra) -> np.ndarray | None:
        f = getattr(self.lib, fname)
        cnt = f(n_rows, pct, *extra, None, 0)
        if cnt < 0:
            return None
        out = np.zeros(max(cnt, 1), dtype=np.uint64)
        got = f(n_rows, pct, *extra, _ptr(out, _u64p), cnt)
        assert got == cnt
        return out[:cnt]

    def count(self, fname: str, n_rows: int, pct: float, *extra) -> int:
        """How many rows the reference's sampler returns (no index list: usable at full table sizes)."""
        return int(getattr(self.lib, fname)(n_rows, pct, *extra, None, 0))

    def idx_memory_stride(self, M, pct, stride_bytes=0): return self._idx("aqo_idx_memory_stride", M, pct, stride_bytes)
    def idx_address_arithmetic(self, M, pct): return self._idx("aqo_idx_address_arithmetic", M, pct)
    def idx_random_pointer(self, N, pct, seed=42): return self._idx("aqo_idx_random_pointer", N, pct, seed)
    def idx_direct_access(self, N, pct): return self._idx("aqo_idx_direct_access", N, pct)
    def idx_optimized_sequential(self, N, pct, seed=42): return self._idx("aqo_idx_optimized_sequential", N, pct, seed)

    def leaf_sizes(self, N):
        self.lib.aqo_leaf_sizes.restype = C.c_int64
        self.lib.aqo_leaf_sizes.argtypes = [C.c_uint64, C.c_void_p, C.c_int64]
        n = self.lib.aqo_leaf_sizes(N, None, 0)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self.lib.aqo_leaf_sizes(N, out.ctypes.data, n)
        return out[:n]

    def uniform_real(self, seed, hi):
        self.lib.aqo_uniform_real.restype = C.c_double
        self.lib.aqo_uniform_real.argtypes = [C.c_uint32, C.c_double]
        return self.lib.aqo_uniform_real(seed, hi)

    def idx_block(self, N, pct, B=1000): return self._idx("aqo_idx_block", N, pct, B)
    def idx_page(self, N, pct, page=4096): return self._idx("aqo_idx_page", N, pct, page)
    def idx_parallel_block(self, N, pct, B=1000, T=4): return self._idx("aqo_idx_parallel_block", N, pct, B, T)
    def idx_optimized_clt(self, N, pct, T=4): return self._idx("aqo_idx_optimized_clt", N, pct, T)
    def idx_fast_pointer(self, N, pct, step_size=2): return self._idx("aqo_idx_fast_pointer", N, pct, step_size)
    def idx_slow_pointer(self, N, pct): return self._idx("aqo_idx_fast_pointer", N, pct, 1)
    def idx_dual_pointer(self, N, pct): return self._idx("aqo_idx_dual_pointer", N, pct)
    def idx_parallel_pointer(self, N, pct, T=4): return self._idx("aqo_idx_parallel_pointer", N, pct, T)

    def idx_adaptive_block(self, rows, pct, min_block=500, max_block=2000):
        f = self.lib.aqo_idx_adaptive_block
        cnt = f(rows.ctypes.data, len(rows), pct, min_block, max_block, None, None, 0)
        if cnt < 0:
            return None
        out = np.zeros(max(cnt, 1), dtype=np.uint64)
        f(rows.ctypes.data, len(rows), pct, min_block, max_block, None, _ptr(out, _u64p), cnt)
        return out[:cnt]

    def idx_stratified_block(self, rows, pct, B=1000, strata=4):
        f = self.lib.aqo_idx_stratified_block
        cnt = f(rows.ctypes.data, len(rows), pct, B, strata, None, 0)
        if cnt < 0:
            return None
        out = np.zeros(max(cnt, 1), dtype=np.uint64)
        f(rows.ctypes.data, len(rows), pct, B, strata, _ptr(out, _u64p), cnt)
        return out[:cnt]

    def idx_random_start_stride(self, M, pct, stride_bytes=0, seed=42, start=None):
        sp = None
        if start is not None:
            st = np.array([start], dtype=np.uint64)
            sp = _ptr(st, _u64p)
        f = self.lib.aqo_idx_random_start_stride
        cnt = f(M, pct, stride_bytes, seed, sp, None, 0)
        out = np.zeros(max(cnt, 1), dtype=np.uint64)
        f(M, pct, stride_bytes, seed, sp, _ptr(out, _u64p), cnt)
        return out[:cnt]

    def idx_region_stride(self, M, pct, T=4, seed=42, starts=None, reference_partition=False):
        sp = None
        if starts is not None:
            starts = np.ascontiguousarray(starts, dtype=np.uint64)
            sp = _ptr(starts, _u64p)
        f = self.lib.aqo_idx_region_stride
        cnt = f(M, pct, T, seed, sp, int(reference_partition), None, 0)
        if cnt < 0:
            return None
        out = np.zeros(max(cnt, 1), dtype=np.uint64)
        f(M, pct, T, seed, sp, int(reference_partition), _ptr(out, _u64p), cnt)
        return out[:cnt]

    # ---- reductions ----
    def moments_idx(self, rows: np.ndarray, idx: np.ndarray, where=None) -> Moments:
        idx = np.ascontiguousarray(idx, dtype=np.uint64)
        m = Moments()
        hw, lo, hi = (1, where[0], where[1]) if where else (0, 0.0, 0.0)
        self.lib.aqo_moments_idx(rows.ctypes.data, _ptr(idx, _u64p), len(idx), hw, lo, hi, C.byref(m))
        return m

    def moments_range(self, rows: np.ndarray, lo: int, hi: int, where=None) -> Moments:
        m = Moments()
        hw, a, b = (1, where[0], where[1]) if where else (0, 0.0, 0.0)
        self.lib.aqo_moments_range(rows.ctypes.data, lo, hi, hw, a, b, C.byref(m))
        return m

    def ci_cli(self, agg, N, n, m2, estimate):
        lo, hi = C.c_double(), C.c_double()
        moe = self.lib.aqo_ci_cli(agg, N, n, m2, estimate, C.byref(lo), C.byref(hi))
        return moe, lo.value, hi.value

    # ---- CLT ----
    def clt_plan(self, N, pct, conf=0.95, check_interval=10, T=4):
        p = CltPlan()
        rc = self.lib.aqo_clt_make_plan(N, pct, conf, check_interval, T, C.byref(p))
        return rc, p

    def clt_run(self, rows, pct, conf=0.95, check_interval=10, T=4, e=2.0, R0=None, growth=1,
                want_idx=False):
        N = len(rows)
        R0 = check_interval if R0 is None else R0
        res = CltResult()
        n_idx = C.c_int64(0)
        idx = None
        if want_idx:
            cap = 3 * int(N * pct / 100.0) + 16 * T + 16
            idx = np.zeros(cap, dtype=np.uint64)
            rc = self.lib.aqo_clt_run(rows.ctypes.data, N, pct, conf, check_interval, T, e, R0, growth,
                                      C.byref(res), _ptr(idx, _u64p), cap, C.byref(n_idx))
            assert n_idx.value <= cap
            idx = idx[: n_idx.value]
        else:
            rc = self.lib.aqo_clt_run(rows.ctypes.data, N, pct, conf, check_interval, T, e, R0, growth,
                                      C.byref(res), None, 0, C.byref(n_idx))
        return rc, res, idx

    def clt_round_partial(self, rows_shard, lo, hi, plan, ord_begin, ord_end):
        f, s = Moments(), Moments()
        self.lib.aqo_clt_round_partial(rows_shard.ctypes.data, lo, hi, C.byref(plan), ord_begin, ord_end,
                                       C.byref(f), C.byref(s))
        return f, s

    # ---- files ----
    def file_write(self, path, rows, height=1):
        return self.lib.aqo_file_write(str(path).encode(), rows.ctypes.data, len(rows), height)

    def file_read(self, path, first=0, cap=None):
        n = self.lib.aqo_file_count(str(path).encode())
        if n < 0:
            return None
        cap = n if cap is None else cap
        rows = np.zeros(max(cap, 1), dtype=REC)
        got = self.lib.aqo_file_read(str(path).encode(), rows.ctypes.data, first, cap)
        return rows[:got]

    def mt19937(self, seed, n):
        out = np.zeros(n, dtype=np.uint32)
        self.lib.aqo_mt19937_stream(seed, out.ctypes.data_as(C.POINTER(C.c_uint32)), n)
        return out


# method numbers of oracle/ref_harness.cpp::ref_sample
REF_METHODS = {
    "memory_stride_sample": 1, "optimized_address_arithmetic_sample": 2, "random_pointer_sample": 3,
    "block_sample": 4, "page_sample": 5, "parallel_block_sample": 6, "optimized_clt_sample": 7,
    "clt_validated_dual_pointer_sample": 8, "fast_pointer_sample": 9, "slow_pointer_sample": 10,
    "dual_pointer_sample": 11, "parallel_pointer_sample": 12, "multithreaded_memory_stride_sample": 13,
    "random_start_memory_stride_sample": 14, "signal_based_clt_sample": 15, "sample_records": 16,
    "adaptive_block_sample": 17, "stratified_block_sample": 18, "direct_access_sample": 19, "optimized_sequential_sample": 20,
}


def ref_available() -> bool:
    return REF_SO.exists()


class Ref:
    """The reference's CustomBPlusDB, driven through oracle/ref_harness.cpp."""

    def __init__(self):
        if not REF_SO.exists():
            raise FileNotFoundError(f"{REF_SO} not built (needs /root/reference; run make -C oracle ref)")
        L = self.lib = C.CDLL(str(REF_SO))
        L.ref_create.restype = C.c_void_p
        L.ref_destroy.argtypes = [C.c_void_p]
        L.ref_fill_direct.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.ref_fill_insert.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        for n in ("ref_total_records", "ref_cache_rows", "ref_node_count", "ref_tree_height"):
            getattr(L, n).restype = C.c_uint64
            getattr(L, n).argtypes = [C.c_void_p]
        L.ref_sum_amount.restype = C.c_double
        L.ref_sum_amount.argtypes = [C.c_void_p]
        L.ref_avg_amount.restype = C.c_double
        L.ref_avg_amount.argtypes = [C.c_void_p]
        L.ref_sum_amount_where.restype = C.c_double
        L.ref_sum_amount_where.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.ref_parallel_sum_sample.restype = C.c_double
        L.ref_parallel_sum_sample.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.ref_parallel_avg_sample.restype = C.c_double
        L.ref_parallel_avg_sample.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.ref_parallel_count_sample.restype = C.c_uint64
        L.ref_parallel_count_sample.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.ref_parallel_sum_where_sample.restype = C.c_double
        L.ref_parallel_sum_where_sample.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_int]
        L.ref_fast_aggregated_memory_stride_sum.restype = C.c_double
        L.ref_fast_aggregated_memory_stride_sum.argtypes = [C.c_void_p, C.c_double, C.c_int]
        L.ref_sample.restype = C.c_int64
        L.ref_sample.argtypes = [C.c_void_p, C.c_int] + [C.c_double] * 5
        L.ref_last_ids.restype = C.c_int64
        L.ref_last_ids.argtypes = [C.c_void_p, _i64p, C.c_int64]
        L.ref_last_amounts.restype = C.c_int64
        L.ref_last_amounts.argtypes = [C.c_void_p, _f64p, C.c_int64]
        L.ref_last_rows.restype = C.c_int64
        L.ref_last_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        L.ref_save_to_file.argtypes = [C.c_void_p, C.c_char_p]
        L.ref_sched_confidence.restype = C.c_double
        L.ref_sched_confidence.argtypes = [C.c_double, C.c_uint64]
        L.ref_sched_where.argtypes = [C.c_char_p, _f64p, _f64p]
        self.h = L.ref_create()

    def close(self):
        if self.h:
            self.lib.ref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def fill_direct(self, rows: np.ndarray):
        assert rows.dtype == REC and rows.flags.c_contiguous
        return self.lib.ref_fill_direct(self.h, rows.ctypes.data, len(rows))

    def fill_insert(self, rows: np.ndarray):
        return self.lib.ref_fill_insert(self.h, rows.ctypes.data, len(rows))

    def total_records(self): return int(self.lib.ref_total_records(self.h))
    def cache_rows(self): return int(self.lib.ref_cache_rows(self.h))
    def node_count(self): return int(self.lib.ref_node_count(self.h))
    def tree_height(self): return int(self.lib.ref_tree_height(self.h))
    def sum_amount(self): return self.lib.ref_sum_amount(self.h)
    def avg_amount(self): return self.lib.ref_avg_amount(self.h)
    def sum_amount_where(self, lo, hi): return self.lib.ref_sum_amount_where(self.h, lo, hi)
    def parallel_sum_sample(self, pct, t=4): return self.lib.ref_parallel_sum_sample(self.h, pct, t)
    def parallel_avg_sample(self, pct, t=4): return self.lib.ref_parallel_avg_sample(self.h, pct, t)
    def parallel_count_sample(self, pct, t=4): return int(self.lib.ref_parallel_count_sample(self.h, pct, t))
    def parallel_sum_where_sample(self, lo, hi, pct, t=4):
        return self.lib.ref_parallel_sum_where_sample(self.h, lo, hi, pct, t)
    def fast_aggregated_memory_stride_sum(self, pct, t=4):
        return self.lib.ref_fast_aggregated_memory_stride_sum(self.h, pct, t)

    def leaf_sizes(self):
        """key_count of every leaf, in leaf order (the next_leaf chain)."""
        self.lib.ref_leaf_sizes.restype = C.c_int64
        self.lib.ref_leaf_sizes.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        n = self.lib.ref_leaf_sizes(self.h, None, 0)
        out = np.zeros(max(n, 1), dtype=np.uint32)
        self.lib.ref_leaf_sizes(self.h, out.ctypes.data, n)
        return out[:n]

    def uniform_real(self, seed, hi):
        """std::uniform_real_distribution<double>(0, hi)(std::mt19937(seed)) of this container's libstdc++."""
        self.lib.ref_uniform_real.restype = C.c_double
        self.lib.ref_uniform_real.argtypes = [C.c_uint32, C.c_double]
        return self.lib.ref_uniform_real(seed, hi)

    def sample(self, method: str, pct: float, a=0.0, b=0.0, c=0.0, d=0.0):
        """Run a sampler; returns the ids (np.int64, reference order) or None if it threw."""
        n = self.lib.ref_sample(self.h, REF_METHODS[method], pct, a, b, c, d)
        if n < 0:
            return None
        ids = np.zeros(max(n, 1), dtype=np.int64)
        self.lib.ref_last_ids(self.h, _ptr(ids, _i64p), n)
        return ids[:n]

    def last_amounts(self, n):
        out = np.zeros(max(n, 1), dtype=np.float64)
        self.lib.ref_last_amounts(self.h, _ptr(out, _f64p), n)
        return out[:n]

    def last_rows(self, n):
        out = np.zeros(max(n, 1), dtype=REC)
        self.lib.ref_last_rows(self.h, out.ctypes.data, n)
        return out[:n]

    def save_to_file(self, path): return bool(self.lib.ref_save_to_file(self.h, str(path).encode()))
    def sched_confidence(self, pct, total): return self.lib.ref_sched_confidence(pct, total)

    def sched_where(self, query: str):
        lo, hi = C.c_double(), C.c_double()
        self.lib.ref_sched_where(query.encode(), C.byref(lo), C.byref(hi))
        return lo.value, hi.value
