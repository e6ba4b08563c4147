"""engine.py — thin object layer over the C ABI (include/aqe_hip.h): one Engine = one GPU = one shard.

The Engine owns an ``aqe_ctx``; every number it returns was computed by the HIP kernels behind
``aqe_reduce`` / the plan API.  There is no CPU path here.
"""
from __future__ import annotations

import ctypes as C
import weakref
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _native as nat
from ._native import AqeError, Query, Result  # noqa: F401

#: numpy view of the reference's 32-byte row (custom_bplus_db.hpp:17-27)
RECORD_DTYPE = np.dtype(
    [("id", "<i8"), ("amount", "<f8"), ("region", "<i4"), ("product_id", "<i4"), ("timestamp", "<i8")]
)
assert RECORD_DTYPE.itemsize == 32


class Plan:
    """A planned query on one Engine (aqe_plan): rounds can be enqueued one by one (multi-GPU) or at once."""

    def __init__(self, engine: "Engine", query: Query, families=None, global_samples: int = 0, on_sorted: bool = False):
        self.engine = engine
        self.query = query
        self._h = C.c_void_p()
        if families is None:
            nat.check(nat.lib().aqe_plan_create(engine._h, C.byref(query), C.byref(self._h)), engine._h)
        else:  # the caller's families instead of the sampler's own (aqe_plan_create_families)
            arr = (nat.Family * max(len(families), 1))(*families)
            nat.check(nat.lib().aqe_plan_create_families(engine._h, C.byref(query), arr, len(families), int(global_samples),
                                                         1 if on_sorted else 0, C.byref(self._h)), engine._h)
        engine._plans.add(self)
        r, t = C.c_uint32(), C.c_int32()
        nat.check(nat.lib().aqe_plan_rounds(self._h, C.byref(r), C.byref(t)), engine._h)
        self.rounds, self.has_topup = r.value, bool(t.value)
        n = C.c_uint32()
        nat.check(nat.lib().aqe_plan_totals_len(self._h, C.byref(n)), engine._h)
        self.totals_len = n.value  # 0: no batched (one-collective) form

    def close(self):
        if self._h:
            nat.lib().aqe_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        nat.check(rc, self.engine._h)

    def reset(self, stream: int = 0):
        self._chk(nat.lib().aqe_plan_reset(self._h, C.c_void_p(stream)))

    def enqueue_round(self, r: int, dev_vec_ptr: int, stream: int = 0):
        self._chk(nat.lib().aqe_plan_enqueue_round(self._h, r, C.c_void_p(dev_vec_ptr), C.c_void_p(stream)))

    def enqueue_update(self, r: int, dev_vec_ptr: int, stream: int = 0):
        self._chk(nat.lib().aqe_plan_enqueue_update(self._h, r, C.c_void_p(dev_vec_ptr), C.c_void_p(stream)))

    def enqueue_finalize(self, stream: int = 0):
        self._chk(nat.lib().aqe_plan_enqueue_finalize(self._h, C.c_void_p(stream)))

    def enqueue_sweep_totals(self, dev_totals_ptr: int, stream: int = 0):
        self._chk(nat.lib().aqe_plan_enqueue_sweep_totals(self._h, C.c_void_p(dev_totals_ptr), C.c_void_p(stream)))

    def enqueue_replay(self, dev_totals_ptr: int, stream: int = 0):
        self._chk(nat.lib().aqe_plan_enqueue_replay(self._h, C.c_void_p(dev_totals_ptr), C.c_void_p(stream)))

    def enqueue_all(self, stream: int = 0):
        self._chk(nat.lib().aqe_plan_enqueue_all(self._h, C.c_void_p(stream)))

    def fetch(self, stream: int = 0) -> Result:
        res = Result()
        self._chk(nat.lib().aqe_plan_fetch(self._h, C.byref(res), C.c_void_p(stream)))
        return res

    def set_profiling(self, enable: bool = True):
        self._chk(nat.lib().aqe_plan_set_profiling(self._h, int(enable)))

    def launch_ms(self):
        n = C.c_uint32()
        self._chk(nat.lib().aqe_plan_launch_ms(self._h, None, 0, C.byref(n)))
        buf = (C.c_float * max(n.value, 1))()
        self._chk(nat.lib().aqe_plan_launch_ms(self._h, buf, n.value, C.byref(n)))
        return list(buf[: n.value])

    def launch_samples(self):
        n = C.c_uint32()
        self._chk(nat.lib().aqe_plan_launch_samples(self._h, None, 0, C.byref(n)))
        buf = (C.c_uint64 * max(n.value, 1))()
        self._chk(nat.lib().aqe_plan_launch_samples(self._h, buf, n.value, C.byref(n)))
        return list(buf[: n.value])

    def last_kernel(self) -> int:
        """Which kernel swept the rounds of the most recent execution (nat.KERNEL_*)."""
        k = C.c_int()
        self._chk(nat.lib().aqe_plan_last_kernel(self._h, C.byref(k)))
        return k.value

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        self._chk(nat.lib().aqe_plan_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value


class Batch:
    """Plans of one Engine executed together (aqe_batch).

    Single GPU: ``enqueue_all(stream)`` runs every plan's query in ONE launch (a group of workgroups per plan, each
    with its own monitor wave and should_stop word) and ``fetch()`` returns the results.

    Multi-GPU: the same one launch with the decisions left out (``enqueue_sweeps``: this shard's round totals per
    plan, on the engine's side stream), ``join(stream)`` makes the caller's stream — where the collective is
    issued — wait for it, and ``enqueue_replays`` decides every plan from the reduced totals in one more launch.
    Three host calls per step for the whole batch."""

    def __init__(self, plans):
        self.plans = list(plans)
        self.engine = self.plans[0].engine
        arr = (C.c_void_p * len(self.plans))(*[p._h for p in self.plans])
        self._h = C.c_void_p()
        nat.check(nat.lib().aqe_batch_create(arr, len(self.plans), C.byref(self._h)), self.engine._h)
        self.engine._batches.add(self)

    def close(self):
        if self._h:
            nat.lib().aqe_batch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def enqueue_sweeps(self, dev_totals_ptr: int, row_stride: int):
        nat.check(nat.lib().aqe_batch_enqueue_sweeps(self._h, C.c_void_p(dev_totals_ptr), row_stride), self.engine._h)

    def join(self, stream: int = 0):
        nat.check(nat.lib().aqe_batch_join(self._h, C.c_void_p(stream)), self.engine._h)

    def enqueue_replays(self, dev_totals_ptr: int, row_stride: int, stream: int = 0):
        nat.check(nat.lib().aqe_batch_enqueue_replays(self._h, C.c_void_p(dev_totals_ptr), row_stride, C.c_void_p(stream)), self.engine._h)

    def fetch(self):
        out = (Result * len(self.plans))()
        nat.check(nat.lib().aqe_batch_fetch(self._h, out), self.engine._h)
        return list(out)

    def enqueue_all(self, stream: int = 0):
        nat.check(nat.lib().aqe_batch_enqueue_all(self._h, C.c_void_p(stream)), self.engine._h)

    def set_profiling(self, enable: bool = True):
        nat.check(nat.lib().aqe_batch_set_profiling(self._h, int(enable)), self.engine._h)

    def launch_info(self, timed: bool = True):
        """(milliseconds, rows swept, workgroups) of the most recent one-launch execution (ms None unless profiled)."""
        ms, n, g = C.c_float(), C.c_uint64(), C.c_uint32()
        nat.check(nat.lib().aqe_batch_launch_info(self._h, C.byref(ms) if timed else None, C.byref(n), C.byref(g)), self.engine._h)
        return (ms.value if timed else None), n.value, g.value


class Comm:
    """RCCL communicator behind the C ABI (aqe_comm): in-place f64 all-reduce on the engine's GPU.

    One process per GPU: rank 0 calls ``Comm.unique_id()``, hands the 128 bytes to every rank out of band, and every
    rank constructs ``Comm(engine, id, nranks, rank)``.  One process, several GPUs: ``Comm.create_all(engines)``."""

    def __init__(self, engine: "Engine", unique_id: bytes, nranks: int, rank: int, _handle=None):
        self.engine = engine
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
        else:
            if len(unique_id) != nat.COMM_ID_BYTES:
                raise ValueError("unique_id must be 128 bytes (Comm.unique_id())")
            buf = C.create_string_buffer(bytes(unique_id), nat.COMM_ID_BYTES)
            nat.check(nat.lib().aqe_comm_create(engine._h, buf, nranks, rank, C.byref(self._h)), engine._h)
        n, r = C.c_int(), C.c_int()
        nat.check(nat.lib().aqe_comm_info(self._h, C.byref(n), C.byref(r)), engine._h)
        self.nranks, self.rank = n.value, r.value
        engine._comms.add(self)

    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(nat.COMM_ID_BYTES)
        nat.check(nat.lib().aqe_comm_unique_id(buf), None)
        return buf.raw

    @classmethod
    def create_all(cls, engines):
        arr = (C.c_void_p * len(engines))(*[e._h for e in engines])
        out = (C.c_void_p * len(engines))()
        nat.check(nat.lib().aqe_comm_create_all(arr, len(engines), out), engines[0]._h)
        return [cls(e, b"", 0, 0, _handle=C.c_void_p(h)) for e, h in zip(engines, out)]

    @classmethod
    def over_mailbox(cls, engine: "Engine", mailbox: "Mailbox"):
        """A communicator whose SUM all-reduces go through a connected Mailbox instead of RCCL (aqe_comm_create_mailbox): what
        run_plan / run_batch then use.  Close it before the mailbox."""
        h = C.c_void_p()
        nat.check(nat.lib().aqe_comm_create_mailbox(engine._h, mailbox._h, C.byref(h)), engine._h)
        return cls(engine, b"", 0, 0, _handle=h)

    def close(self):
        if self._h:
            nat.lib().aqe_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def all_reduce_sum(self, dev_ptr: int, count: int, stream: int = 0):
        nat.check(nat.lib().aqe_comm_all_reduce_sum(self._h, C.c_void_p(dev_ptr), count, C.c_void_p(stream)), self.engine._h)

    def all_reduce_max(self, dev_ptr: int, count: int, stream: int = 0):
        nat.check(nat.lib().aqe_comm_all_reduce_max(self._h, C.c_void_p(dev_ptr), count, C.c_void_p(stream)), self.engine._h)

    def run_plan(self, plan: "Plan", dev_vec_ptr: int, stream: int = 0) -> Result:
        """aqe_plan_run_sharded: the whole query over the ranks, host side in C."""
        res = Result()
        nat.check(nat.lib().aqe_plan_run_sharded(plan._h, self._h, C.c_void_p(dev_vec_ptr), C.c_void_p(stream), C.byref(res)), self.engine._h)
        return res

    def run_batch(self, batch: "Batch", dev_totals_ptr: int, row_stride: int, stream: int = 0):
        """aqe_batch_run_sharded: sweeps, join, ONE all-reduce, replays — one host call per step (asynchronous)."""
        nat.check(nat.lib().aqe_batch_run_sharded(batch._h, self._h, C.c_void_p(dev_totals_ptr), row_stride, len(batch.plans),
                                                  C.c_void_p(stream)), self.engine._h)


class Mailbox:
    """The one-shot peer-mapped all-reduce (aqe_mailbox): every rank writes its vector into every peer's mailbox and adds up
    what arrived, in ONE single-workgroup launch — for the moment vectors of the multi-GPU path, where a collective is pure
    latency.  One process per GPU: ``mb = Mailbox(engine, nranks, rank)``, exchange ``mb.handle()`` (64 bytes) in rank order,
    ``mb.connect(handles)``.  One process, several GPUs: ``Mailbox.connect_local(mailboxes)``."""

    def __init__(self, engine: "Engine", nranks: int, rank: int):
        self.engine, self.nranks, self.rank = engine, nranks, rank
        self._h = C.c_void_p()
        nat.check(nat.lib().aqe_mailbox_create(engine._h, nranks, rank, C.byref(self._h)), engine._h)
        engine._comms.add(self)

    def handle(self) -> bytes:
        buf = C.create_string_buffer(nat.MAILBOX_HANDLE_BYTES)
        nat.check(nat.lib().aqe_mailbox_handle(self._h, buf), self.engine._h)
        return buf.raw

    def connect(self, handles) -> None:
        blob = b"".join(bytes(h) for h in handles)
        if len(blob) != self.nranks * nat.MAILBOX_HANDLE_BYTES:
            raise ValueError("one 64-byte handle per rank, in rank order")
        nat.check(nat.lib().aqe_mailbox_connect(self._h, C.create_string_buffer(blob, len(blob))), self.engine._h)

    @staticmethod
    def connect_local(mailboxes) -> None:
        arr = (C.c_void_p * len(mailboxes))(*[m._h for m in mailboxes])
        nat.check(nat.lib().aqe_mailbox_connect_local(arr, len(mailboxes)), mailboxes[0].engine._h)

    def all_reduce_sum(self, dev_ptr: int, count: int, stream: int = 0):
        nat.check(nat.lib().aqe_mailbox_all_reduce_sum(self._h, C.c_void_p(dev_ptr), count, C.c_void_p(stream)), self.engine._h)

    def late_ranks(self) -> int:
        """Bit r set: rank r did not show up within the bound in some all-reduce (whose vector was then left untouched)."""
        v = C.c_uint32()
        nat.check(nat.lib().aqe_mailbox_status(self._h, C.byref(v)), self.engine._h)
        return v.value

    def close(self):
        if self._h:
            nat.lib().aqe_mailbox_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One GPU context holding one shard [shard_lo, shard_lo+local_rows) of a table of global_rows rows."""

    def __init__(self, device_id: int = 0):
        self._h = C.c_void_p()
        rc = nat.lib().aqe_create(device_id, C.byref(self._h))
        if rc != nat.OK:
            nat.check(rc, None)
        self._keepalive = None  # tensors adopted through attach_device
        # what was made on this context and is still alive: a context goes only after them (batches before their plans),
        # whatever order the caller — or a garbage collector after a failed test — lets go of things in
        self._plans, self._batches, self._comms = weakref.WeakSet(), weakref.WeakSet(), weakref.WeakSet()

    # -- lifecycle --
    def close(self):
        if self._h:
            for group in (self._batches, self._comms, self._plans):
                for obj in list(group):
                    obj.close()
            nat.lib().aqe_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _chk(self, rc):
        nat.check(rc, self._h)

    # -- staging --
    def stage_records(self, rows: np.ndarray, shard_lo: int = 0, n_global: Optional[int] = None, keep_aos: bool = True):
        rows = np.ascontiguousarray(rows, dtype=RECORD_DTYPE)
        n_global = len(rows) if n_global is None else n_global
        self._chk(nat.lib().aqe_stage_records(self._h, rows.ctypes.data, len(rows), shard_lo, n_global,
                                              nat.STAGE_KEEP_AOS if keep_aos else 0))

    def stage_file(self, path, shard_lo: int = 0, n_local: int = 0, keep_aos: bool = True):
        self._chk(nat.lib().aqe_stage_file(self._h, str(path).encode(), shard_lo, n_local,
                                           nat.STAGE_KEEP_AOS if keep_aos else 0))

    def stage_stats(self) -> nat.StageStats:
        """Where the time of the most recent stage_records / stage_file went (aqe_last_stage_stats)."""
        st = nat.StageStats()
        self._chk(nat.lib().aqe_last_stage_stats(self._h, C.byref(st)))
        return st

    def save_file(self, path):
        self._chk(nat.lib().aqe_save_file(self._h, str(path).encode()))

    def generate_synthetic(self, n_local: int, shard_lo: int = 0, n_global: Optional[int] = None, seed: int = 42,
                           keep_aos: bool = False):
        n_global = n_local if n_global is None else n_global
        self._chk(nat.lib().aqe_generate_synthetic(self._h, n_local, shard_lo, n_global, seed,
                                                   nat.STAGE_KEEP_AOS if keep_aos else 0))

    def attach_device(self, amount_ptr: int, n_local: int, shard_lo: int, n_global: int, shift: float,
                      aos_ptr: int = 0, keepalive=None):
        self._chk(nat.lib().aqe_attach_device(self._h, C.c_void_p(amount_ptr), C.c_void_p(aos_ptr), n_local,
                                              shard_lo, n_global, shift))
        self._keepalive = keepalive

    def set_shift(self, shift: float):
        self._chk(nat.lib().aqe_set_shift(self._h, shift))

    def release_table(self):
        self._chk(nat.lib().aqe_release_table(self._h))
        self._keepalive = None

    def info(self) -> nat.TableInfo:
        t = nat.TableInfo()
        self._chk(nat.lib().aqe_table_info_get(self._h, C.byref(t)))
        return t

    def key_range_rows(self, id_min: int, id_max: int) -> Tuple[int, int]:
        """Row window [lo, hi) of `id BETWEEN id_min AND id_max` (rows are in ascending-id leaf order)."""
        lo, hi = C.c_uint64(), C.c_uint64()
        self._chk(nat.lib().aqe_key_range_rows(self._h, id_min, id_max, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def key_range_counts(self, id_min: int, id_max: int) -> Tuple[int, int]:
        """(rows of THIS shard with id < id_min, rows with id <= id_max): summed over the shards they are the global row window."""
        lo, hi = C.c_uint64(), C.c_uint64()
        self._chk(nat.lib().aqe_key_range_counts(self._h, id_min, id_max, C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    # -- hot path --
    def reduce(self, query: Query) -> Result:
        res = Result()
        self._chk(nat.lib().aqe_reduce(self._h, C.byref(query), C.byref(res)))
        return res

    def reduce_grouped(self, query: Query, group_column: int, max_groups: int = 1024):
        """GROUP BY region / product_id with a per-group interval (executor.cpp:202-321): list of GroupResult,
        ascending key, only keys with at least one sampled row."""
        out = self._group_buf(max_groups)  # (kept with the engine: allocating and zeroing 72 KB per call cost more than the launch)
        n = C.c_uint32()
        self._chk(nat.lib().aqe_reduce_grouped(self._h, C.byref(query), int(group_column), out, max_groups, C.byref(n)))
        return list((nat.GroupResult * n.value).from_buffer_copy(out)) if n.value else []  # (one copy out of the kept buffer)

    def _group_buf(self, max_groups: int):
        buf = getattr(self, "_grp_buf", None)
        if buf is None or len(buf) < max_groups:
            buf = self._grp_buf = (nat.GroupResult * max_groups)()
        return buf

    # multi-GPU form of reduce_grouped: key range -> (all-reduce MIN/MAX) -> bins -> (all-reduce SUM) -> finish
    def group_key_range(self, group_column: int):
        lo, hi = C.c_int32(), C.c_int32()
        self._chk(nat.lib().aqe_group_key_range(self._h, int(group_column), C.byref(lo), C.byref(hi)))
        return lo.value, hi.value

    def grouped_enqueue_bins(self, query: Query, group_column: int, key_min: int, nbins: int, dev_bins_ptr: int, stream: int = 0):
        self._chk(nat.lib().aqe_grouped_enqueue_bins(self._h, C.byref(query), int(group_column), int(key_min), int(nbins),
                                                     C.c_void_p(dev_bins_ptr), C.c_void_p(stream)))

    def grouped_finish(self, query: Query, key_min: int, nbins: int, dev_bins_ptr: int, stream: int = 0, max_groups: int = 1024):
        out = (nat.GroupResult * max_groups)()
        n = C.c_uint32()
        self._chk(nat.lib().aqe_grouped_finish(self._h, C.byref(query), int(key_min), int(nbins), C.c_void_p(dev_bins_ptr),
                                               C.c_void_p(stream), out, max_groups, C.byref(n)))
        return list(out[: n.value])

    def gather(self, query: Query) -> np.ndarray:
        """Rows of the record-returning sampler, as a RECORD_DTYPE array."""
        n = C.c_uint64()
        rc = nat.lib().aqe_gather(self._h, C.byref(query), None, 0, C.byref(n))
        if rc not in (nat.OK, nat.ERR_CAPACITY):
            self._chk(rc)
        out = np.zeros(max(n.value, 1), dtype=RECORD_DTYPE)
        if n.value:
            self._chk(nat.lib().aqe_gather(self._h, C.byref(query), out.ctypes.data, n.value, C.byref(n)))
        return out[: n.value]

    def plan(self, query: Query) -> Plan:
        return Plan(self, query)

    def plan_families(self, query: Query, families, global_samples: int, on_sorted: bool = False) -> Plan:
        """A single-round plan over the given families (nat.Family): rows of the table in global numbering, or — on_sorted —
        positions in this engine's amount-sorted column.  `query` supplies aggregate, estimators, sample_percent, WHERE."""
        return Plan(self, query, families=list(families), global_samples=global_samples, on_sorted=on_sorted)

    # ---- what the ranks of a sharded table exchange before the variance-aware samplers can plan (distributed.py) ----
    def zone_moments(self) -> np.ndarray:
        """[10, 3] float64: (rows, sum, sum of squares) of the rows of each of adaptive_block_sample's ten zones held here."""
        out = np.zeros(30, dtype=np.float64)
        self._chk(nat.lib().aqe_zone_moments(self._h, out.ctypes.data_as(C.POINTER(C.c_double))))
        return out.reshape(10, 3)

    def set_zone_variances(self, var10) -> None:
        v = np.ascontiguousarray(var10, dtype=np.float64)
        if v.shape != (10,):
            raise ValueError("ten zone variances")
        self._chk(nat.lib().aqe_set_zone_variances(self._h, v.ctypes.data_as(C.POINTER(C.c_double))))

    def sorted_counts(self, values):
        """(rows with amount < v, rows with amount <= v) of this engine's shard for every v of `values` (uint64 arrays)."""
        v = np.ascontiguousarray(values, dtype=np.float64).ravel()
        lt = np.zeros(len(v), dtype=np.uint64)
        le = np.zeros(len(v), dtype=np.uint64)
        if len(v):
            self._chk(nat.lib().aqe_sorted_counts(self._h, v.ctypes.data_as(C.POINTER(C.c_double)), len(v),
                                                  lt.ctypes.data_as(C.POINTER(C.c_uint64)), le.ctypes.data_as(C.POINTER(C.c_uint64))))
        return lt, le


def make_query(method: int, sample_percent: float = 10.0, agg: int = nat.SUM, convention: int = nat.EST_CLI,
               where: Optional[Tuple[float, float]] = None, **kw) -> Query:
    """aqe_query with the reference's defaults (bindings.cpp:56-101) plus overrides."""
    rows = kw.pop("rows", None)
    q = nat.default_query(method=method, sample_percent=float(sample_percent), agg=agg, convention=convention, **kw)
    if rows is not None:
        q.row_lo, q.row_hi = int(rows[0]), int(rows[1])
    if where is not None:
        q.has_where, q.where_min, q.where_max = 1, float(where[0]), float(where[1])
    return q
