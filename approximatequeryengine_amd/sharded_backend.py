"""sharded_backend.py — the CustomBPlusDB query API (aqe_backend.py; reference: bindings.cpp:42-101) over a table SHARDED by
row region across the ranks of a torch.distributed group: one process per GPU, every rank holds rows
[⌊g·N/G⌋, ⌊(g+1)·N/G⌋) of the leaf-order array in its own HBM, every query is one sweep per rank plus ONE all-reduce of the
moment vectors (distributed.ShardedQuery), and every rank returns the same answer — the one a single GPU holding the whole table
gives.  What the reference does inside one process with a mutex, `future.get()` and a CAS loop (custom_bplus_db.cpp:948-951,
966-967, 2031-2036) over its region partition (custom_bplus_db.cpp:1903-1921).

    torchrun --nproc-per-node 8 -m approximatequeryengine_amd.cli "SELECT AVG(amount) FROM sales" --db sales.db --error 0.01
    # (spell --error / --sample out under torchrun: its own parser takes `--e`, `--s` for abbreviations of its options)
    # or, in a program every rank runs:
    db = ShardedBPlusDB()                 # joins the default process group; GPU = LOCAL_RANK
    db.open_database("sales.db")          # every rank stages only its region of the file
    r = db.approx("AVG", method="clt", error_percent=0.01)     # same ApproxResult on every rank

All of approx() (`id_between` included: the key bounds are counted per shard and summed), approx_batch(), approx_group_by(), the
exact aggregates and the fused `parallel_*_sample` entry points work;
the two samplers that need a fact about the whole table (adaptive_block: zone variances; stratified_block: a global sort)
agree on it over the group first (distributed.sharded_adaptive_plan / sharded_stratified_plan).  The record-RETURNING samplers
stay per GPU: a sharded table has no single process to hand a list of records to."""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np

from . import _native as nat
from .aqe_backend import ApproxResult, CustomBPlusDB, GroupEstimate, _AGG
from .distributed import (MOMENT_VEC, ShardedBatch, ShardedQuery, shard_bounds, sharded_adaptive_plan, sharded_group_by,
                          sharded_stratified_plan, torch_all_reduce, torch_host_all_reduce)
from .engine import RECORD_DTYPE, Batch, Engine, make_query


class ShardedBPlusDB(CustomBPlusDB):
    """CustomBPlusDB over the ranks of a process group.  Every method below is COLLECTIVE: every rank calls it with the same
    arguments, in the same order."""

    def __init__(self, *, device_id: Optional[int] = None, group=None, keep_rows_on_device: bool = True, collective: str = "torch"):
        import torch
        import torch.distributed as dist
        if not dist.is_initialized():
            raise RuntimeError("ShardedBPlusDB needs an initialised torch.distributed process group (one process per GPU)")
        super().__init__(device_id=int(os.environ.get("LOCAL_RANK", "0")) if device_id is None else device_id, keep_rows_on_device=keep_rows_on_device)
        self._group, self._rank, self._world = group, dist.get_rank(group), dist.get_world_size(group)
        torch.cuda.set_device(self._device_id)
        self._side = torch.cuda.Stream(device=self._device_id)
        self._dev = torch.device("cuda", self._device_id)
        self._ar_sum, self._ar_max = torch_all_reduce(group), torch_all_reduce(group, op="max")
        self._host_ar = torch_host_all_reduce(group, device=self._dev if dist.get_backend(group) == "nccl" else None)
        self._mailbox = None
        if collective == "mailbox" and self._world > 1:  # peer-mapped one-launch all-reduce for the moment vectors (aqe_mailbox_*)
            self._want_mailbox = True
        elif collective == "torch" or self._world == 1:
            self._want_mailbox = False
        else:
            raise ValueError("collective must be 'torch' or 'mailbox'")
        self._plans = {}
        self._vec = None
        self._bins = None
        self._lo = self._hi = 0

    # ---- the table: every rank stages its own region ----
    def _adopt(self, n_global: int):
        if n_global < self._world:
            raise ValueError(f"a table of {n_global} rows cannot be sharded over {self._world} ranks")
        self._drop_plans()
        self._n = n_global
        self._lo, self._hi = shard_bounds(n_global, self._world, self._rank)
        if self._engine is None:
            self._engine = Engine(self._device_id)

    def open_database(self, db_path: str) -> bool:
        return self.load_from_file(db_path)

    def load_from_file(self, file_path: str) -> bool:
        """Every rank stages rows [lo, hi) of the file (offset 24 + 32·lo) — no rank reads the whole table."""
        n = nat.C.c_uint64()
        if nat.lib().aqe_file_rows(str(file_path).encode(), nat.C.byref(n)) != nat.OK:
            return False
        self._adopt(int(n.value))
        self._engine.stage_file(file_path, shard_lo=self._lo, n_local=self._hi - self._lo, keep_aos=self._keep_aos)
        self._finish_staging()
        return True

    def insert_array(self, rows: np.ndarray) -> bool:
        """The WHOLE table in leaf order (ascending id) on every rank; each keeps its region.  (For tables that do not fit one
        host, write the file once and let every rank open it.)"""
        rows = np.ascontiguousarray(rows, dtype=RECORD_DTYPE)
        if len(rows) > 1 and np.any(rows["id"][1:] < rows["id"][:-1]):
            rows = rows[np.lexsort((-np.arange(len(rows)), rows["id"]))]  # leaf order: ascending id, equal ids newest first
        self._adopt(len(rows))
        self._engine.stage_records(rows[self._lo:self._hi], shard_lo=self._lo, n_global=len(rows), keep_aos=self._keep_aos)
        head = rows[: min(len(rows), 1024)]["amount"]
        self._engine.set_shift(float(np.add.reduce(head) / len(head)))  # the table's head, the same on every rank
        self._finish_staging()
        return True

    def generate_synthetic(self, n_global: int, seed: int = 42) -> bool:
        """The seeded synthetic table of SURVEY 8d, each rank generating its own region on its GPU."""
        self._adopt(int(n_global))
        self._engine.generate_synthetic(self._hi - self._lo, shard_lo=self._lo, n_global=int(n_global), seed=seed, keep_aos=self._keep_aos)
        self._finish_staging()
        return True

    def insert_record(self, record) -> bool:
        raise NotImplementedError("a sharded table is loaded in bulk: insert_array, generate_synthetic or open_database")

    insert_batch = insert_record

    def save_to_file(self, file_path: str) -> bool:
        raise NotImplementedError("a sharded table is not written back: every rank holds only its region")

    def close_database(self) -> None:
        self._drop_plans()
        if self._mailbox is not None:
            import torch.distributed as dist
            dist.barrier(group=self._group)  # nobody is still writing into a mailbox that is about to go
            self._mailbox.close()
            self._mailbox = None
        if self._engine is not None:
            self._engine.close()
            self._engine = None

    def _finish_staging(self):
        self._dirty = False
        if self._want_mailbox and self._mailbox is None:
            from .distributed import mailbox_from_torch_group
            self._mailbox = mailbox_from_torch_group(self._engine, self._group)

    def _drop_plans(self):
        for p in self._plans.values():
            p.close()
        self._plans = {}

    def _eng(self) -> Engine:
        if self._engine is None or self._n == 0:
            raise RuntimeError("no table loaded")
        return self._engine

    def shard(self):
        """(lo, hi): the global rows this rank holds."""
        return self._lo, self._hi

    # ---- one query over all ranks ----
    def _collective(self, numel: int):
        if self._mailbox is not None and numel <= nat.MAILBOX_MAX_DOUBLES:
            from .distributed import mailbox_all_reduce
            return mailbox_all_reduce(self._mailbox, self._side.cuda_stream)
        return self._ar_sum

    def _plan_for(self, q):
        key = bytes(q)
        p = self._plans.get(key)
        if p is None:
            if q.method == nat.M_ADAPTIVE_BLOCK:
                p = sharded_adaptive_plan(self._engine, q, self._host_ar)
            elif q.method == nat.M_STRATIFIED_BLOCK:
                p = sharded_stratified_plan(self._engine, q, self._host_ar, self._rank, self._world)
            else:
                p = self._engine.plan(q)
            if len(self._plans) >= 64:
                self._drop_plans()
            self._plans[key] = p
        return p

    def _buffer(self, need: int):
        import torch
        if self._vec is None or self._vec.numel() < need:
            self._vec = torch.zeros(max(need, 1024), dtype=torch.float64, device=self._dev)
        return self._vec

    def _reduce(self, q) -> nat.Result:
        import torch
        self._eng()
        plan = self._plan_for(q)
        need = max(MOMENT_VEC, plan.totals_len)
        with torch.cuda.stream(self._side):
            vec = self._buffer(need)
            return ShardedQuery(plan, vec, self._collective(need), stream=self._side.cuda_stream).run()

    def _gather(self, q, as_array: bool):
        raise NotImplementedError("record-returning samplers are per GPU: a sharded table has no single process to hand the rows to "
                                  "(use approx() / approx_batch() / approx_group_by(), or CustomBPlusDB on one GPU)")

    def _key_window(self, id_min: int, id_max: int):
        """B+-tree key bounds over shards: ids ascend over the whole table, so every rank counts its rows below id_min / up to
        id_max (aqe_key_range_counts) and ONE all-reduce SUM of the two counts is the global row window."""
        below, upto = self._eng().key_range_counts(id_min, id_max)
        w = self._host_ar(np.array([float(below), float(upto)]))
        return int(w[0]), int(w[1])

    def _random_cpp(self, agg, sample_percent, seed, where=None) -> nat.Result:
        # (the reference draws from std::random_device; here every rank must draw the SAME sample: rank 0's seed)
        if seed is None:
            import torch.distributed as dist
            box = [int.from_bytes(os.urandom(8), "little") if self._rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(self._group, 0) if self._group is not None else 0, group=self._group)
            seed = box[0]
        return super()._random_cpp(agg, sample_percent, seed, where)

    def approx_batch(self, queries: "List[dict]") -> "List[ApproxResult]":
        """Several APPROX queries with ONE collective for all those that have the batched form (multi-round CLT queries: their
        sweeps are one launch, their round totals one all-reduce, their replays one launch); the others one after another."""
        import torch
        self._eng()
        specs = [dict(kw) for kw in queries]
        qs = [self._approx_query(**kw) for kw in specs]
        plans = [self._plan_for(q) for q in qs]
        out: "List[Optional[ApproxResult]]" = [None] * len(qs)
        fused = [i for i, p in enumerate(plans) if p.totals_len > 0]
        with torch.cuda.stream(self._side):
            if len(fused) >= 2:
                ps = [plans[i] for i in fused]
                width = max(p.totals_len for p in ps)
                buf = self._buffer(len(ps) * width)[: len(ps) * width].view(len(ps), width)
                batch = Batch(ps)
                try:
                    res = ShardedBatch(ps, buf, self._collective(len(ps) * width), stream=self._side.cuda_stream, batch=batch).run()
                finally:
                    batch.close()
                for i, r in zip(fused, res):
                    out[i] = ApproxResult(r, specs[i].get("method", "stride"))
            for i, q in enumerate(qs):
                if out[i] is None:
                    out[i] = ApproxResult(self._reduce(q), specs[i].get("method", "stride"))
        for r in out:
            if r.visited == 0:
                raise RuntimeError("No samples collected")
        return out

    def approx_group_by(self, agg: str, group_by: str = "region", sample_percent: float = 10.0, method: str = "rowid",
                        where=None, block_size: int = 1000) -> "dict[str, GroupEstimate]":
        """GROUP BY over all ranks: the key range is agreed (one MAX all-reduce), every rank bins the part of the sample inside
        its region, ONE all-reduce SUM merges the bins (distributed.sharded_group_by)."""
        import torch
        col = {"region": nat.GROUP_REGION, "product_id": nat.GROUP_PRODUCT}[group_by.strip().lower()]
        m = {"rowid": nat.M_ROWID_MOD, "stride": nat.M_MEMORY_STRIDE, "block": nat.M_BLOCK, "page": nat.M_PAGE, "exact": nat.M_EXACT}[method]
        self._eng()
        bs = 4096 if (method == "page" and block_size == 1000) else block_size
        q = make_query(m, sample_percent, agg=_AGG[agg.upper()], where=where, block_size=int(bs))
        with torch.cuda.stream(self._side):
            if self._bins is None:
                self._bins = torch.zeros(4 * 4096, dtype=torch.float64, device=self._dev)
            groups = sharded_group_by(self._engine, q, col, self._bins, self._ar_sum, self._ar_max, stream=self._side.cuda_stream)
        return {str(r.key): GroupEstimate(r) for r in groups}
