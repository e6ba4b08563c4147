"""aqe_backend — drop-in Python mirror of the reference's pybind11 module of the same name
(/root/reference/src/aqe_backend/bindings/bindings.cpp:10-137) for the sampled SUM/AVG/COUNT path.

Same class names, method names, argument order and defaults; the numbers come from the HIP kernels behind
the C ABI (include/aqe_hip.h), never from Python arithmetic and never from a CPU fallback.  Differences
from the reference are deliberate and listed in INTEGRATION.md; the important ones:

* ``open_database`` / ``load_from_file`` work (the reference dead-locks in load_from_file,
  custom_bplus_db.cpp:689 -> 165);
* the flat row array the samplers see is always the whole table (the reference's cache goes stale
  between multiples of 1000 inserts, custom_bplus_db.cpp:188-191);
* the CLT monitor is round-synchronous and deterministic instead of racing std::async workers;
* samplers the reference seeds from std::random_device take a ``seed`` and are reproducible;
* fused ``approx_sum / approx_avg / approx_count`` return the aggregate + interval without materialising
  Python ``Record`` objects (the reference reduces in Python, enhanced_aqe_cli.py:189-200).

Tree-walking samplers that §8 of SURVEY.md rules out raise NotImplementedError (the two the reference CLI routes small
tables to — direct_access_sample, optimized_sequential_sample — are in: their row lists follow from the leaf shape the
reference builds from ascending inserts).
"""
from __future__ import annotations

import ctypes as C
import enum
import os
import time
from datetime import timedelta
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from .engine import RECORD_DTYPE, Engine, make_query

__all__ = ["Record", "CustomBPlusDB", "CustomApproximateScheduler", "CustomValidationResult",
           "CustomApproximationStatus", "ApproxResult", "BenchmarkResults", "GroupEstimate"]


class Record:
    """bindings.cpp:14-20 — read/write fields id, amount, region, product_id, timestamp."""
    __slots__ = ("id", "amount", "region", "product_id", "timestamp")

    def __init__(self, id: int = 0, amount: float = 0.0, region: int = 0, product_id: int = 0, timestamp: int = 0):
        self.id, self.amount, self.region, self.product_id, self.timestamp = id, amount, region, product_id, timestamp

    def __repr__(self):
        return f"Record(id={self.id}, amount={self.amount}, region={self.region}, product_id={self.product_id}, timestamp={self.timestamp})"

    def __eq__(self, other):
        return isinstance(other, Record) and all(getattr(self, k) == getattr(other, k) for k in self.__slots__)


class CustomApproximationStatus(enum.Enum):
    """custom_scheduler.hpp:8-13"""
    STABLE = 0
    DRIFTING = 1
    INSUFFICIENT_DATA = 2
    ERROR = 3


class CustomValidationResult:
    """custom_scheduler.hpp:15-22; computation_time is a datetime.timedelta as pybind11/chrono.h yields."""
    __slots__ = ("value", "status", "confidence_level", "error_margin", "samples_used", "computation_time")

    def __init__(self, value=0.0, status=CustomApproximationStatus.ERROR, confidence_level=0.0, error_margin=100.0,
                 samples_used=0, computation_time=timedelta(0)):
        self.value, self.status, self.confidence_level = value, status, confidence_level
        self.error_margin, self.samples_used, self.computation_time = error_margin, samples_used, computation_time

    def __repr__(self):
        return (f"CustomValidationResult(value={self.value}, status={self.status.name}, confidence_level={self.confidence_level}, "
                f"error_margin={self.error_margin}, samples_used={self.samples_used}, computation_time={self.computation_time})")


class BenchmarkResults:
    """custom_scheduler.hpp:73-82 (the reference never registers this type with pybind11)."""
    __slots__ = ("exact_value", "approximate_value", "exact_time_ms", "approximate_time_ms", "speedup",
                 "error_percentage", "threads_used", "sample_percentage")

    def __init__(self, **kw):
        for k in self.__slots__:
            setattr(self, k, kw.get(k, 0))


class ApproxResult:
    """What a caller of the reference computes from a sample (value, 95 % interval, moments), produced on
    the GPU in one call."""
    __slots__ = ("value", "ci_lower", "ci_upper", "margin", "n", "visited", "sum", "sumsq", "mean", "m2", "converged",
                 "rounds", "topup", "kernel_ms", "bytes_algorithmic", "achieved_GBps", "method")

    def __init__(self, res: nat.Result, method: str):
        for k in ("value", "ci_lower", "ci_upper", "margin", "n", "visited", "sum", "sumsq", "mean", "m2", "converged",
                  "rounds", "topup", "kernel_ms", "bytes_algorithmic"):
            setattr(self, k, getattr(res, k))
        self.achieved_GBps = (res.bytes_algorithmic / (res.kernel_ms * 1e-3) / 1e9) if res.kernel_ms > 0 else 0.0
        self.method = method

    def __repr__(self):
        return (f"ApproxResult(value={self.value!r}, ci=({self.ci_lower!r}, {self.ci_upper!r}), n={self.n}, "
                f"converged={self.converged}, rounds={self.rounds}, method={self.method!r})")


_AGG = {"SUM": nat.SUM, "AVG": nat.AVG, "COUNT": nat.COUNT}
_OUT_OF_SCOPE = (
    "index_based_sample", "node_skip_sample", "balanced_tree_sample",
    "byte_offset_sample", "random_start_nth_sample", "address_arithmetic_sample",
    "signal_based_clt_sample",
)


class GroupEstimate:
    """One group of approx_group_by: executor.h's QueryResult {value, ci_lower, ci_upper} plus the moments behind it."""
    __slots__ = ("value", "ci_lower", "ci_upper", "n", "sum", "mean")

    def __init__(self, r):
        self.value, self.ci_lower, self.ci_upper = r.value, r.ci_lower, r.ci_upper
        self.n, self.sum, self.mean = int(r.n), r.sum, r.mean

    def __repr__(self):
        return f"GroupEstimate(value={self.value:.6g}, ci=[{self.ci_lower:.6g}, {self.ci_upper:.6g}], n={self.n})"

    def __iter__(self):  # unpacks like the reference's (value, ci_lower, ci_upper)
        return iter((self.value, self.ci_lower, self.ci_upper))


def parse_where(query: str) -> Optional[Tuple[float, float]]:
    """The amount range of a query's WHERE clause as the scheduler reads it (custom_scheduler.cpp:277-294), or None."""
    lo, hi = C.c_double(), C.c_double()
    return (lo.value, hi.value) if nat.lib().aqe_parse_where(query.encode(), C.byref(lo), C.byref(hi)) else None


def _records(arr: np.ndarray) -> List[Record]:
    """numpy rows -> list[Record], what pybind11's list_caster gives the reference's callers."""
    return [Record(int(i), float(a), int(r), int(p), int(t))
            for i, a, r, p, t in zip(arr["id"].tolist(), arr["amount"].tolist(), arr["region"].tolist(),
                                     arr["product_id"].tolist(), arr["timestamp"].tolist())]


class CustomBPlusDB:
    """bindings.cpp:42-101.  Rows live in a host staging buffer until the first query, then in HBM."""

    def __init__(self, *, device_id: int = 0, keep_rows_on_device: bool = True):
        self._device_id = device_id
        self._keep_aos = keep_rows_on_device
        self._engine: Optional[Engine] = None
        self._rows = np.zeros(1024, dtype=RECORD_DTYPE)
        self._n = 0
        self._sorted = True
        self._last_id = None
        self._dirty = True
        self._path = ""

    # ---- lifecycle (custom_bplus_db.cpp:135-162) ----
    def create_database(self, db_path: str) -> bool:
        self._path = str(db_path)
        self._n, self._sorted, self._last_id, self._dirty = 0, True, None, True
        return True

    def open_database(self, db_path: str) -> bool:
        return self.load_from_file(db_path)

    def close_database(self) -> None:
        if self._path:  # the reference auto-saves when a path was set (custom_bplus_db.cpp:157-162)
            self.save_to_file(self._path)
        if self._engine is not None:
            self._engine.close()
            self._engine = None
            self._dirty = True

    def __del__(self):
        try:
            if self._engine is not None:
                self._engine.close()
        except Exception:
            pass

    # ---- rows ----
    def _reserve(self, extra: int):
        need = self._n + extra
        if need > len(self._rows):
            cap = max(need, 2 * len(self._rows))
            grown = np.zeros(cap, dtype=RECORD_DTYPE)
            grown[: self._n] = self._rows[: self._n]
            self._rows = grown

    def insert_record(self, record: Record) -> bool:
        self._reserve(1)
        self._rows[self._n] = (record.id, record.amount, record.region, record.product_id, record.timestamp)
        if self._last_id is not None and record.id < self._last_id:
            self._sorted = False
        self._last_id = record.id if self._last_id is None else max(self._last_id, record.id)
        self._n += 1
        self._dirty = True
        return True

    def insert_batch(self, records: Iterable[Record]) -> bool:
        for r in records:
            self.insert_record(r)
        return True

    def insert_array(self, rows: np.ndarray) -> bool:
        """Bulk insert of a RECORD_DTYPE array (no per-row Python objects; replaces the reference's
        O(N^2/1000) insert path, custom_bplus_db.cpp:188-191)."""
        rows = np.ascontiguousarray(rows, dtype=RECORD_DTYPE)
        if len(rows) == 0:
            return True
        self._reserve(len(rows))
        self._rows[self._n: self._n + len(rows)] = rows
        ids = rows["id"]
        if (self._last_id is not None and ids[0] < self._last_id) or (len(ids) > 1 and np.any(ids[1:] < ids[:-1])):
            self._sorted = False
        self._last_id = int(ids.max()) if self._last_id is None else max(self._last_id, int(ids.max()))
        self._n += len(rows)
        self._dirty = True
        return True

    def _leaf_order(self) -> np.ndarray:
        """Rows in B+-tree leaf order: ascending id; equal ids newest-first (lower_bound insert,
        custom_bplus_db.cpp:31-37)."""
        rows = self._rows[: self._n]
        if not self._sorted:
            seq = np.arange(self._n)
            order = np.lexsort((-seq, rows["id"]))
            rows = rows[order]
            self._rows[: self._n] = rows
            self._sorted = True
        return rows

    def _eng(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self._device_id)
            self._dirty = True
        if self._dirty:
            self._engine.stage_records(self._leaf_order(), keep_aos=self._keep_aos)
            self._dirty = False
        return self._engine

    # ---- files (custom_bplus_db.cpp:665-711) ----
    def save_to_file(self, file_path: str) -> bool:
        try:
            rows = self._leaf_order()
            height = 1
            cap = 254
            while self._n > cap:
                height, cap = height + 1, cap * 128
            with open(file_path, "wb") as f:
                f.write(np.array([self._n, height, self._n], dtype="<u8").tobytes())
                f.write(rows.tobytes())
            return True
        except OSError:
            return False

    def load_from_file(self, file_path: str) -> bool:
        try:
            size = os.path.getsize(file_path)
            if size < 24:
                return False
            hdr = np.fromfile(file_path, dtype="<u8", count=3)
            count = int(hdr[2])
            if 24 + 32 * count > size:
                return False
            rows = np.memmap(file_path, dtype=RECORD_DTYPE, mode="r", offset=24, shape=(count,)) if count else \
                np.zeros(0, dtype=RECORD_DTYPE)
        except OSError:
            return False
        self._n, self._sorted, self._last_id = 0, True, None
        self._rows = np.zeros(max(count, 1024), dtype=RECORD_DTYPE)
        self.insert_array(np.asarray(rows))
        return True

    # ---- statistics ----
    def get_total_records(self) -> int:
        return self._n

    def get_node_count(self) -> int:
        return self._n // 255 + 1  # custom_bplus_db.cpp:654-658

    def get_tree_height(self) -> int:
        height, cap = 1, 254
        while self._n > cap:
            height, cap = height + 1, cap * 128
        return height

    # ---- exact (custom_bplus_db.cpp:242-274) ----
    def _reduce(self, q) -> nat.Result:
        return self._eng().reduce(q)

    def sum_amount(self) -> float:
        return self._reduce(make_query(nat.M_EXACT, 100.0, agg=nat.SUM)).value if self._n else 0.0

    def avg_amount(self) -> float:
        return self._reduce(make_query(nat.M_EXACT, 100.0, agg=nat.AVG)).value if self._n else 0.0

    def count_records(self) -> int:
        return self._n

    def sum_amount_where(self, min_amount: float, max_amount: float) -> float:
        return self._reduce(make_query(nat.M_EXACT, 100.0, where=(min_amount, max_amount))).value if self._n else 0.0

    # ---- record-returning samplers (bindings.cpp:50-101) ----
    def _gather(self, q, as_array: bool):
        if self._n == 0:
            return np.zeros(0, dtype=RECORD_DTYPE) if as_array else []
        arr = self._eng().gather(q)
        return arr if as_array else _records(arr)

    def memory_stride_sample(self, sample_percent, stride_bytes=0, *, as_array=False):
        return self._gather(make_query(nat.M_MEMORY_STRIDE, sample_percent, stride_bytes=int(stride_bytes)), as_array)

    def random_start_memory_stride_sample(self, sample_percent, stride_bytes=0, *, seed=None, as_array=False):
        """custom_bplus_db.cpp:1838-1878; the reference draws the start from std::random_device, here it is seeded."""
        seed = int.from_bytes(os.urandom(8), "little") if seed is None else int(seed)
        return self._gather(make_query(nat.M_RANDOM_START_STRIDE, sample_percent, stride_bytes=int(stride_bytes), seed=seed), as_array)

    def optimized_address_arithmetic_sample(self, sample_percent, *, as_array=False):
        return self._gather(make_query(nat.M_ADDRESS_ARITHMETIC, sample_percent), as_array)

    def direct_access_sample(self, sample_percent, *, as_array=False):
        """custom_bplus_db.cpp:584-644 — what the reference CLI takes for 10 k < N <= 50 k rows: ~10 % of the B+ tree's leaves
        at a fixed node step, evenly spaced records in each.  The leaves are those the reference builds from ascending inserts
        (127 rows each, the last 128 ... 254); rows in the reference's order, duplicates included."""
        return self._gather(make_query(nat.M_DIRECT_ACCESS, sample_percent), as_array)

    def optimized_sequential_sample(self, sample_percent, *, seed=None, as_array=False):
        """custom_bplus_db.cpp:366-428 — the reference CLI's sampler for N <= 10 k rows: systematic, step 100 / pct from a
        random start (std::random_device there; seeded here)."""
        seed = int.from_bytes(os.urandom(4), "little") if seed is None else int(seed) & 0xFFFFFFFF
        return self._gather(make_query(nat.M_OPTIMIZED_SEQUENTIAL, sample_percent, seed=seed), as_array)

    def random_pointer_sample(self, sample_percent, seed=42, *, as_array=False):
        return self._gather(make_query(nat.M_RANDOM_POINTER, sample_percent, seed=int(seed) & 0xFFFFFFFF), as_array)

    def sample_records(self, sample_percent, *, seed=None, as_array=False):
        """custom_bplus_db.cpp:345-363: a uniform sample without replacement of floor(N*pct/100) rows.  The
        reference shuffles with std::random_device; here the draw is the seeded mt19937 one."""
        if sample_percent >= 100.0:
            return self._gather(make_query(nat.M_MEMORY_STRIDE, 100.0), as_array)
        seed = int.from_bytes(os.urandom(4), "little") if seed is None else int(seed)
        return self.random_pointer_sample(sample_percent, seed, as_array=as_array)

    def block_sample(self, sample_percent, block_size=1000, *, as_array=False):
        return self._gather(make_query(nat.M_BLOCK, sample_percent, block_size=int(block_size)), as_array)

    def page_sample(self, sample_percent, page_size=4096, *, as_array=False):
        return self._gather(make_query(nat.M_PAGE, sample_percent, block_size=int(page_size)), as_array)

    def parallel_block_sample(self, sample_percent, block_size=1000, num_threads=4, *, as_array=False):
        return self._gather(make_query(nat.M_PARALLEL_BLOCK, sample_percent, block_size=int(block_size),
                                       num_threads=int(num_threads)), as_array)

    def adaptive_block_sample(self, sample_percent, min_block_size=500, max_block_size=2000, *, as_array=False):
        """custom_bplus_db.cpp:1273-1329 (the ten zone variances come from a device pre-pass, cached per table)."""
        return self._gather(make_query(nat.M_ADAPTIVE_BLOCK, sample_percent, block_size=int(min_block_size),
                                       block_size_max=int(max_block_size)), as_array)

    def stratified_block_sample(self, sample_percent, block_size=1000, strata_count=4, *, as_array=False):
        """custom_bplus_db.cpp:1331-1379 (the amount column is sorted once on the device, cached per table)."""
        return self._gather(make_query(nat.M_STRATIFIED_BLOCK, sample_percent, block_size=int(block_size),
                                       num_threads=int(strata_count)), as_array)

    def optimized_clt_sample(self, sample_percent, confidence_level=0.95, check_interval=20, num_threads=4,
                             max_error_percent=2.0, *, as_array=False):
        return self._gather(make_query(nat.M_OPTIMIZED_CLT, sample_percent, confidence_level=confidence_level,
                                       check_interval=int(check_interval), num_threads=int(num_threads),
                                       max_error_percent=max_error_percent), as_array)

    def clt_validated_dual_pointer_sample(self, sample_percent, confidence_level=0.95, check_interval=10, num_threads=4,
                                          max_error_percent=2.0, *, round0=0, growth=1, as_array=False):
        return self._gather(self._clt_query(sample_percent, confidence_level, check_interval, num_threads,
                                            max_error_percent, round0, growth, nat.AVG), as_array)

    def fast_pointer_sample(self, sample_percent, step_size=2, *, as_array=False):
        return self._gather(make_query(nat.M_FAST_POINTER, sample_percent, step_size=int(step_size)), as_array)

    def slow_pointer_sample(self, sample_percent, *, as_array=False):
        return self._gather(make_query(nat.M_SLOW_POINTER, sample_percent), as_array)

    def dual_pointer_sample(self, sample_percent, *, as_array=False):
        return self._gather(make_query(nat.M_DUAL_POINTER, sample_percent), as_array)

    def parallel_pointer_sample(self, sample_percent, num_threads=4, *, as_array=False):
        return self._gather(make_query(nat.M_PARALLEL_POINTER, sample_percent, num_threads=int(num_threads)), as_array)

    def multithreaded_memory_stride_sample(self, sample_percent, num_threads=4, *, seed=42, as_array=False):
        return self._gather(make_query(nat.M_REGION_STRIDE, sample_percent, num_threads=int(num_threads), seed=int(seed)),
                            as_array)

    # ---- C++-side reducers (custom_bplus_db.cpp:276-343, 1962-2048) ----
    def fast_aggregated_memory_stride_sum(self, sample_percent, num_threads=4, *, seed=42) -> float:
        if self._n == 0:
            return 0.0
        return self._reduce(make_query(nat.M_REGION_STRIDE, sample_percent, convention=nat.EST_RAW,
                                       num_threads=int(num_threads), seed=int(seed))).value

    def _random_cpp(self, agg, sample_percent, seed, where=None) -> nat.Result:
        # sample_records (DB.cpp:345-363) shuffles with std::random_device: there is nothing to match bit for bit, so
        # the sample is drawn ON THE DEVICE (AQE_M_RANDOM_DEVICE: a keyed bijection of the rows, no host index list)
        seed = int.from_bytes(os.urandom(8), "little") if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        return self._reduce(make_query(nat.M_RANDOM_DEVICE, sample_percent, agg=agg, convention=nat.EST_CPP, seed=seed,
                                       where=where))

    def parallel_sum_sample(self, sample_percent, num_threads=4, *, seed=None) -> float:
        return self._random_cpp(nat.SUM, sample_percent, seed).value if self._n else 0.0

    def parallel_avg_sample(self, sample_percent, num_threads=4, *, seed=None) -> float:
        return self._random_cpp(nat.AVG, sample_percent, seed).value if self._n else 0.0

    def parallel_count_sample(self, sample_percent, num_threads=4, *, seed=None) -> int:
        return int(self._random_cpp(nat.COUNT, sample_percent, seed).value) if self._n else 0

    def parallel_sum_where_sample(self, min_amount, max_amount, sample_percent, num_threads=4, *, seed=None) -> float:
        return self._random_cpp(nat.SUM, sample_percent, seed, where=(min_amount, max_amount)).value if self._n else 0.0

    # ---- fused aggregate entry points (the point of the GPU path) ----
    def _clt_query(self, pct, conf, ci, T, e, round0, growth, agg):
        # The reference checks every `check_interval` samples per worker (round0 = 0, growth = 1 reproduces that
        # cadence round for round).  On a big table that is hundreds of thousands of decision points; past 4096 of
        # them the cadence keeps its first check and doubles from there — it can only stop later than the reference
        # would (same error test at the stop), and the planner's 2^20-round limit is never hit.
        if int(round0) == 0 and int(growth) <= 1 and int(ci) > 0 and int(T) > 0:
            per_worker = self._n * float(pct) / 100.0 / max(1, int(T) // 2)
            if per_worker / int(ci) > 4096:
                growth = 2
        return make_query(nat.M_CLT_DUAL_POINTER, pct, agg=agg, confidence_level=conf, check_interval=int(ci),
                          num_threads=int(T), max_error_percent=e, clt_round0=int(round0), clt_growth=int(growth))

    def approx(self, agg: str, method: str = "stride", sample_percent: float = 10.0, error_percent: Optional[float] = None,
               where: Optional[Tuple[float, float]] = None, seed: int = 42, num_threads: int = 4, block_size: int = 1000,
               confidence_level: float = 0.95, check_interval: int = 10, round0: int = 4096, growth: int = 4,
               convention: str = "cli", id_between: Optional[Tuple[int, int]] = None) -> ApproxResult:
        """APPROX <agg>(amount): method in {"stride","random","random_device","block","page","parallel_block","region","clt",
        "exact","adaptive_block","stratified_block"}.  "random" is random_pointer_sample(seed) bit for bit (host mt19937 +
        Lemire index list, DB.cpp:856-882); "random_device" draws a simple random sample on the device (no index list).
        ``error_percent`` (CLT) is in percent, as the reference CLI's --e (enhanced_aqe_cli.py:414-415); the
        sample percentage then follows enhanced_aqe_cli.py:243-250."""
        q = self._approx_query(agg, method, sample_percent, error_percent, where, seed, num_threads, block_size, confidence_level,
                               check_interval, round0, growth, convention, id_between)
        res = self._reduce(q)
        if res.visited == 0:
            raise RuntimeError("No samples collected")
        return ApproxResult(res, method)

    def _approx_query(self, agg, method="stride", sample_percent=10.0, error_percent=None, where=None, seed=42, num_threads=4,
                      block_size=1000, confidence_level=0.95, check_interval=10, round0=4096, growth=4, convention="cli",
                      id_between=None):
        """The aqe_query behind approx(...)."""
        a = _AGG[agg.upper()]
        conv = {"cli": nat.EST_CLI, "cpp": nat.EST_CPP, "raw": nat.EST_RAW}[convention]
        if self._n == 0:
            raise RuntimeError("No samples collected")  # enhanced_aqe_cli.py:226-228
        rows = None
        if id_between is not None:  # B+-tree key bounds -> row window (the pruning search_range never got, DB.hpp:45)
            rows = self._key_window(int(id_between[0]), int(id_between[1]))
            if rows[1] <= rows[0]:
                raise RuntimeError("No samples collected")
        if method == "clt":
            # the CLT monitor returns the reference's own estimate (CLI:262-291) over the whole table: it has no WHERE
            # form (planner.cpp), no alternative convention and no seed — asking for one is an error, not a no-op
            if where is not None:
                raise ValueError("method='clt' has no WHERE form (clt_validated_dual_pointer_sample samples the whole table, DB.cpp:885-1043)")
            if convention != "cli":
                raise ValueError("method='clt' reports the CLI estimate (enhanced_aqe_cli.py:262-291): convention must be 'cli'")
            e = 2.0 if error_percent is None else float(error_percent)
            pct = nat.lib().aqe_error_to_sample_percent(e)
            q = self._clt_query(pct, confidence_level, check_interval, num_threads, e, round0, growth, a)
            if rows:
                q.row_lo, q.row_hi = rows
        else:
            m = {"stride": nat.M_MEMORY_STRIDE, "random": nat.M_RANDOM_POINTER, "random_device": nat.M_RANDOM_DEVICE, "block": nat.M_BLOCK, "page": nat.M_PAGE,
                 "direct_access": nat.M_DIRECT_ACCESS, "sequential": nat.M_OPTIMIZED_SEQUENTIAL,
                 "parallel_block": nat.M_PARALLEL_BLOCK, "region": nat.M_REGION_STRIDE, "exact": nat.M_EXACT,
                 "adaptive_block": nat.M_ADAPTIVE_BLOCK, "stratified_block": nat.M_STRATIFIED_BLOCK}[method]
            if method == "adaptive_block" and block_size == 1000:
                block_size = 500  # the reference's min_block_size default (bindings.cpp:79-80)
            bs = 4096 if (method == "page" and block_size == 1000) else block_size
            q = make_query(m, sample_percent, agg=a, convention=conv, where=where, seed=int(seed),
                           num_threads=int(num_threads), block_size=int(bs), rows=rows)
        return q

    def _key_window(self, id_min: int, id_max: int) -> Tuple[int, int]:
        return self._eng().key_range_rows(id_min, id_max)

    def approx_batch(self, queries: "List[dict]") -> "List[ApproxResult]":
        """Several APPROX queries in ONE launch (aqe_batch_enqueue_all: a group of workgroups, a monitor wave and a
        should_stop word per query): each entry is the keyword dictionary approx() takes, e.g.
        ``[{"agg": "AVG", "method": "clt", "error_percent": 0.01}, {"agg": "SUM", "method": "block", "sample_percent": 1, "where": (250, 750)}]``.
        What replaces the reference's thread creation per call (custom_bplus_db.cpp:918-1029) when queries arrive in
        batches; the seeded samplers that need a host index list ("random") run on their own."""
        from .engine import Batch
        eng = self._eng()
        specs = [dict(kw) for kw in queries]
        qs = [self._approx_query(**kw) for kw in specs]
        out: "List[Optional[ApproxResult]]" = [None] * len(qs)
        fused = [i for i, kw in enumerate(specs) if kw.get("method", "stride") not in ("random", "random_device", "direct_access", "sequential")]
        for i in set(range(len(qs))) - set(fused):
            out[i] = ApproxResult(self._reduce(qs[i]), specs[i].get("method", "stride"))
        if fused:
            plans = [eng.plan(qs[i]) for i in fused]
            batch = None
            try:
                batch = Batch(plans)
                batch.enqueue_all(0)
                for i, r in zip(fused, batch.fetch()):
                    out[i] = ApproxResult(r, specs[i].get("method", "stride"))
            finally:
                if batch is not None:
                    batch.close()
                for p in plans:
                    p.close()
        for r in out:
            if r is None or r.visited == 0:
                raise RuntimeError("No samples collected")
        return out

    def approx_group_by(self, agg: str, group_by: str = "region", sample_percent: float = 10.0, method: str = "rowid",
                        where: Optional[Tuple[float, float]] = None, block_size: int = 1000) -> "dict[str, GroupEstimate]":
        """APPROX <agg>(amount) ... GROUP BY region | product_id with a 95 % interval per group: the reference's
        execute_query_groupby_with_ci (executor.cpp:202-321; GroupResultWithCI = map<string, {value, ci_lower,
        ci_upper}>) in one sweep.  method "rowid" is that function's own sample (rowid % (100 / sample_percent) == 0);
        "stride", "block", "page" and "exact" group the CustomBPlusDB samplers the same way.
        SUM is sum * 100/pct (the reference reports mean * 100/pct under that name; GroupEstimate.mean has the mean)."""
        col = {"region": nat.GROUP_REGION, "product_id": nat.GROUP_PRODUCT}[group_by.strip().lower()]
        m = {"rowid": nat.M_ROWID_MOD, "stride": nat.M_MEMORY_STRIDE, "block": nat.M_BLOCK, "page": nat.M_PAGE, "exact": nat.M_EXACT}[method]
        if self._n == 0:
            return {}
        bs = 4096 if (method == "page" and block_size == 1000) else block_size
        q = make_query(m, sample_percent, agg=_AGG[agg.upper()], where=where, block_size=int(bs))
        return {str(r.key): GroupEstimate(r) for r in self._eng().reduce_grouped(q, col)}

    def approx_sum(self, **kw) -> ApproxResult:
        return self.approx("SUM", **kw)

    def approx_avg(self, **kw) -> ApproxResult:
        return self.approx("AVG", **kw)

    def approx_count(self, **kw) -> ApproxResult:
        return self.approx("COUNT", **kw)


def _not_in_scope(name):
    def f(self, *a, **k):
        raise NotImplementedError(
            f"{name}: tree-walking / sort-based sampler outside the accelerated path (SURVEY.md §8); "
            "use memory_stride_sample, block_sample, random_pointer_sample or the approx_* entry points")
    f.__name__ = name
    return f


for _n in _OUT_OF_SCOPE:
    setattr(CustomBPlusDB, _n, _not_in_scope(_n))


class CustomApproximateScheduler:
    """bindings.cpp:103-123 / custom_scheduler.cpp — the query façade over a CustomBPlusDB."""

    def __init__(self, error_threshold: float = 0.05, *, device_id: int = 0, seed: Optional[int] = None, db: "Optional[CustomBPlusDB]" = None):
        # db: the table to schedule over instead of a new one on `device_id` — e.g. a sharded_backend.ShardedBPlusDB, which makes
        # every execute_* call a collective over the ranks of its process group (pass a `seed` then: every rank must draw the same
        # sample; without one the sharded table agrees on rank 0's)
        self._db = CustomBPlusDB(device_id=device_id) if db is None else db
        self.error_threshold = error_threshold
        self._seed = seed
        self._queries = 0

    def create_database(self, db_path): return self._db.create_database(db_path)
    def open_database(self, db_path): return self._db.open_database(db_path)
    def close_database(self): self._db.close_database()

    def insert_record(self, id, amount, region, product_id, timestamp) -> bool:
        return self._db.insert_record(Record(id, amount, region, product_id, timestamp))

    def insert_batch(self, records) -> bool:
        return self._db.insert_batch(sorted(records, key=lambda r: r.id))  # custom_bplus_db.cpp:196-208

    def insert_array(self, rows) -> bool:
        return self._db.insert_array(rows)

    def _next_seed(self):
        self._queries += 1
        return None if self._seed is None else (self._seed + self._queries) & 0xFFFFFFFF

    def _approx(self, fn, sample_percent) -> CustomValidationResult:
        t0 = time.perf_counter()
        r = CustomValidationResult()
        total = self._db.get_total_records()
        try:
            r.value = fn()
            r.status = CustomApproximationStatus.STABLE
            r.confidence_level = nat.lib().aqe_confidence_heuristic(sample_percent, total)  # custom_scheduler.cpp:296-305
            r.error_margin = sample_percent / 100.0
            r.samples_used = int(total * sample_percent / 100.0)
        except Exception:  # the façade swallows exceptions (custom_scheduler.cpp:74-77)
            r.value = 0.0
            r.status = CustomApproximationStatus.ERROR
        r.computation_time = timedelta(milliseconds=int((time.perf_counter() - t0) * 1000))
        return r

    def execute_sum_query(self, query: str, sample_percent: float = 10.0, num_threads: int = 4) -> CustomValidationResult:
        import ctypes as C
        lo, hi = C.c_double(), C.c_double()
        has = nat.lib().aqe_parse_where(query.encode(), C.byref(lo), C.byref(hi))  # custom_scheduler.cpp:277-294
        seed = self._next_seed()
        if has:
            return self._approx(lambda: self._db.parallel_sum_where_sample(lo.value, hi.value, sample_percent, num_threads, seed=seed), sample_percent)
        return self._approx(lambda: self._db.parallel_sum_sample(sample_percent, num_threads, seed=seed), sample_percent)

    def execute_avg_query(self, query: str, sample_percent: float = 10.0, num_threads: int = 4) -> CustomValidationResult:
        seed = self._next_seed()
        return self._approx(lambda: self._db.parallel_avg_sample(sample_percent, num_threads, seed=seed), sample_percent)

    def execute_count_query(self, query: str, sample_percent: float = 10.0, num_threads: int = 4) -> CustomValidationResult:
        seed = self._next_seed()
        return self._approx(lambda: float(self._db.parallel_count_sample(sample_percent, num_threads, seed=seed)), sample_percent)

    def _exact(self, fn) -> CustomValidationResult:
        t0 = time.perf_counter()
        r = CustomValidationResult(status=CustomApproximationStatus.STABLE, confidence_level=1.0, error_margin=0.0,
                                   samples_used=self._db.get_total_records())
        try:
            r.value = fn()
        except Exception:
            r.value, r.status = 0.0, CustomApproximationStatus.ERROR
        r.computation_time = timedelta(milliseconds=int((time.perf_counter() - t0) * 1000))
        return r

    def execute_exact_sum(self): return self._exact(self._db.sum_amount)
    def execute_exact_avg(self): return self._exact(self._db.avg_amount)
    def execute_exact_count(self): return self._exact(lambda: float(self._db.count_records()))

    def benchmark_query(self, query_type: str, sample_percent: float = 10.0, num_threads: int = 4) -> BenchmarkResults:
        qt = query_type if query_type in ("SUM", "AVG", "COUNT") else "SUM"  # custom_scheduler.cpp:215-229
        exact = {"SUM": self.execute_exact_sum, "AVG": self.execute_exact_avg, "COUNT": self.execute_exact_count}[qt]()
        approx = {"SUM": self.execute_sum_query, "AVG": self.execute_avg_query, "COUNT": self.execute_count_query}[qt](
            f"SELECT {qt}(amount)", sample_percent, num_threads)
        et = exact.computation_time.total_seconds() * 1e3
        at = approx.computation_time.total_seconds() * 1e3
        err = abs(exact.value - approx.value) / abs(exact.value) * 100.0 if exact.value != 0 else 0.0
        return BenchmarkResults(exact_value=exact.value, approximate_value=approx.value, exact_time_ms=et,
                                approximate_time_ms=at, speedup=(et / at if at > 0 else float("inf")),
                                error_percentage=err, threads_used=num_threads, sample_percentage=sample_percent)

    def get_total_records(self): return self._db.get_total_records()
    def get_tree_height(self): return self._db.get_tree_height()
    def get_database_size_mb(self): return self._db.get_total_records() * 32 / (1024.0 * 1024.0)
