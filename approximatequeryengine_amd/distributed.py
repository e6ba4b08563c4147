"""distributed.py — one process per GPU; the table is sharded by contiguous row region and every
convergence step costs exactly one all-reduce of the moment vector (n, Σ(x-c), Σ(x-c)²)×{leader, others}.

The reference merges its workers through a mutex-guarded vector, a CAS loop on atomic<double> and an
atomic<bool> stop flag (custom_bplus_db.cpp:948-951, 966-967, 2031-2036).  Here each rank sweeps the
part of every worker's progression that falls inside its shard (the families are clipped on the host,
planner.cpp), RCCL sums the AQE_MOMENT_VEC doubles over xGMI, and every rank folds the same reduced
vector with the same device kernel — so every rank takes the same stop decision and no broadcast of
should_stop is needed.  A round enqueued after the stop is a device-side no-op on every rank.

This module is pure orchestration: it never touches the numbers.  The object it drives only has to
look like ``engine.Plan`` (rounds, has_topup, reset, enqueue_round, enqueue_update, enqueue_finalize,
fetch), which is how the CPU ``gloo`` tests exercise it with an oracle-backed stand-in.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

MOMENT_VEC = 8


def _stream_for(stream: int, tensor) -> int:
    """The stream a sharded object enqueues on.  A null handle means "the engine's own stream" to the C ABI, which is
    NOT ordered against the stream the framework's collective runs on: for device tensors 0 is therefore resolved to
    torch's current stream, and refused when that is the legacy default stream (whose handle is also 0)."""
    if stream or not getattr(tensor, "is_cuda", False):
        return stream
    import torch
    s = torch.cuda.current_stream(tensor.device).cuda_stream
    if not s:
        raise ValueError("run under `with torch.cuda.stream(side):` or pass stream=side.cuda_stream: the legacy default "
                         "stream cannot be handed to the library, and its own stream is not ordered against the collective")
    return s


class _torch_on:
    """Makes `stream` (a raw handle) torch's current stream for the block, so that what the orchestration does THROUGH torch —
    zeroing the moment vector, a torch.distributed collective — is ordered with the library's launches on that stream even when
    the caller passed the handle explicitly and did not switch torch to it.  A no-op for host tensors (the CPU tests)."""

    def __init__(self, stream: int, tensor):
        self.ctx = None
        if stream and getattr(tensor, "is_cuda", False):
            import torch
            if torch.cuda.current_stream(tensor.device).cuda_stream != stream:
                self.ctx = torch.cuda.stream(torch.cuda.ExternalStream(stream, device=tensor.device))

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
        return False


def shard_bounds(n_global: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Proper prefix partition of the flat row array: rank g owns [⌊g·N/G⌋, ⌊(g+1)·N/G⌋).
    (The reference's own region split, custom_bplus_db.cpp:1903-1921, overlaps by a row and can drop
    the last N mod T rows; that defect is not reproduced.)"""
    return (rank * n_global) // world_size, ((rank + 1) * n_global) // world_size


class ShardedQuery:
    """Runs one planned query across the ranks of a process group.

    plan        an object with the ``engine.Plan`` interface, planned over THIS rank's shard
    vec         a float64 tensor on the plan's device, at least max(MOMENT_VEC, plan.totals_len) long
    all_reduce  callable(tensor) -> None performing an in-place SUM over the group
    stream      raw stream handle passed through to the plan; 0 = torch's current stream for a device buffer (the
                legacy default stream is refused: run under ``with torch.cuda.stream(side)``)
    batched     True: ONE collective per query — every round is swept speculatively in one launch, the
                per-round totals are all-reduced once, and the stop rules are replayed on the reduced totals
                (identical answer; the rounds after the stop are swept for nothing, which on a 10 M-row shard
                costs less than a single extra collective).  The reference's top-up (too few rows collected
                at the stop) is not speculated: when the fetched result says it is due, ``run`` finishes
                with the stepwise top-up step (one more collective, rare).  False: one collective per
                convergence step, nothing swept past the stop.  None: batched when the plan offers it.
    """

    def __init__(self, plan, vec, all_reduce: Callable, stream: int = 0, batched: Optional[bool] = None):
        self.plan = plan
        self.vec = vec
        self.all_reduce = all_reduce
        self.stream = _stream_for(stream, vec)
        can = getattr(plan, "totals_len", 0) > 0
        self.batched = can if batched is None else (batched and can)
        need = plan.totals_len if self.batched else MOMENT_VEC
        if vec.numel() < need:
            raise ValueError(f"moment buffer holds {vec.numel()} doubles, the plan needs {need}")

    def enqueue(self) -> None:
        with _torch_on(self.stream, self.vec):
            self._enqueue()

    def _enqueue(self) -> None:
        p, v, s = self.plan, self.vec, self.stream
        if self.batched:
            t = v[: p.totals_len]
            p.enqueue_sweep_totals(t.data_ptr(), s)  # this shard's total of every round, one launch
            self.all_reduce(t)                       # ONE collective per query
            p.enqueue_replay(t.data_ptr(), s)        # stop rules + estimate, on the device
            return
        p.reset(s)
        steps = p.rounds + (1 if p.has_topup else 0)
        t = v[:MOMENT_VEC]
        for r in range(steps):
            t.zero_()                            # a launch that leaves early writes nothing
            p.enqueue_round(r, t.data_ptr(), s)  # this shard's partial (n, Σd, Σd²)
            self.all_reduce(t)                   # ONE collective per convergence step
            p.enqueue_update(r, t.data_ptr(), s) # fold + CLT rules + should_stop, on the device
        p.enqueue_finalize(s)

    def enqueue_topup(self) -> None:
        """The stepwise top-up step (device-gated): sweep, one collective, fold, estimate."""
        p, s = self.plan, self.stream
        with _torch_on(s, self.vec):
            t = self.vec[:MOMENT_VEC]
            t.zero_()
            p.enqueue_round(p.rounds, t.data_ptr(), s)
            self.all_reduce(t)
            p.enqueue_update(p.rounds, t.data_ptr(), s)
            p.enqueue_finalize(s)

    def run(self):
        self.enqueue()
        res = self.plan.fetch(self.stream)
        if self.batched and _pending(res):  # every rank reads the same mark: every rank comes here
            self.enqueue_topup()
            res = self.plan.fetch(self.stream)
        return res


def _pending(res) -> bool:
    return bool(res["topup_pending"] if isinstance(res, dict) else getattr(res, "topup_pending", 0))


class ShardedBatch:
    """B independent queries per collective: each query's round totals land in its own row of one buffer and a
    single all-reduce serves them all (xGMI collectives are latency-bound at this size: 8 queries x 5 rounds x 64 B
    cost the same ~tens of microseconds as one).  All plans must offer the batched form.

    batch (optional): an ``engine.Batch`` over the same plans.  The sweeps then run on the batch's own side
    streams — one query's hand-off tail overlaps the next query's sweep, as in the single-GPU form — the
    collective (issued on ``stream``) waits for all of them, every replay goes back to its side stream, and the
    host pays two calls per step instead of two per query."""

    def __init__(self, plans, buf, all_reduce: Callable, stream: int = 0, batch=None):
        if any(getattr(p, "totals_len", 0) == 0 for p in plans):
            raise ValueError("every plan of a ShardedBatch needs a batched (totals) form")
        self.plans, self.buf, self.all_reduce, self.stream, self.batch = list(plans), buf, all_reduce, _stream_for(stream, buf), batch
        self.width = max(p.totals_len for p in plans)
        if buf.dim() != 2 or buf.shape[0] < len(self.plans) or buf.shape[1] < self.width or not buf.is_contiguous():
            raise ValueError("buffer must be a contiguous [len(plans), >= totals_len] float64 tensor")

    def enqueue_sweeps(self) -> None:
        """First half of a step: every plan's sweep (this shard's round totals into the plan's row of the buffer)."""
        if self.batch is not None:
            self.batch.enqueue_sweeps(self.buf.data_ptr(), self.buf.shape[1])
            return
        for i, p in enumerate(self.plans):
            p.enqueue_sweep_totals(self.buf[i].data_ptr(), self.stream)

    def finish(self) -> None:
        """Second half: the ONE collective of the batch, then every plan's replay."""
        s, ptr, stride = self.stream, self.buf.data_ptr(), self.buf.shape[1]
        with _torch_on(s, self.buf):
            if self.batch is not None:
                self.batch.join(s)
                self.all_reduce(self.buf)
                self.batch.enqueue_replays(ptr, stride, s)
                return
            self.all_reduce(self.buf)
            for i, p in enumerate(self.plans):
                p.enqueue_replay(self.buf[i].data_ptr(), s)

    def enqueue(self) -> None:
        self.enqueue_sweeps()
        self.finish()

    def fetch(self):
        return self.batch.fetch() if self.batch is not None else [p.fetch(self.stream) for p in self.plans]

    def run(self):
        self.enqueue()
        out = self.fetch()
        due = [i for i, r in enumerate(out) if _pending(r)]
        if due:  # the rare top-ups of the batch share one more collective (same marks on every rank)
            s = self.stream
            with _torch_on(s, self.buf):
                vecs = self.buf.view(-1)[: len(self.plans) * MOMENT_VEC].view(len(self.plans), MOMENT_VEC)  # contiguous
                vecs.zero_()
                for i in due:
                    self.plans[i].enqueue_round(self.plans[i].rounds, vecs[i].data_ptr(), s)
                self.all_reduce(vecs)
                for i in due:
                    self.plans[i].enqueue_update(self.plans[i].rounds, vecs[i].data_ptr(), s)
                    self.plans[i].enqueue_finalize(s)
                    out[i] = self.plans[i].fetch(s)
        return out


class PipelinedBatches:
    """Two (or more) ShardedBatch objects over the same engine, software-pipelined: a step enqueues the sweeps of
    one batch and only then finishes the previous step's batch, so that batch's collective (tens of microseconds
    of latency on xGMI, during which its GPU would idle) runs under the next batch's sweeps.  Same queries, same
    answers; ``flush`` finishes the step still in flight."""

    def __init__(self, batches):
        self.batches = list(batches)
        self._k = 0
        self._pending = None

    def enqueue(self):
        """One step.  Returns the batch whose collective and replays were just enqueued (its results may be fetched
        now: the fetch waits for them while the next batch's sweeps run), or None on the first step."""
        cur = self.batches[self._k % len(self.batches)]
        self._k += 1
        cur.enqueue_sweeps()
        done = self._pending
        if done is not None:
            done.finish()
        self._pending = cur
        return done

    def flush(self):
        """Finishes the step still in flight; returns its batch (or None)."""
        done = self._pending
        if done is not None:
            done.finish()
            self._pending = None
        return done

    def fetch(self):
        self.flush()
        return [r for b in self.batches for r in b.fetch()]


def sharded_group_by(engine, query, group_column: int, bins, all_reduce_sum: Callable, all_reduce_max: Callable, stream: int = 0):
    """GROUP BY across the ranks of a process group (engine.Engine interface): the per-key bins (n, S - c n, Q,
    visited) are additive, so every rank bins the part of the sample inside its shard over the same key range and
    ONE all-reduce SUM merges them; the key range itself is agreed first (one MAX all-reduce of [-min, max]).

    bins            float64 tensor on the engine's device with room for 4 * (number of distinct keys) doubles
    all_reduce_max  callable(tensor) -> None, in-place MAX over the group (``torch_all_reduce(op="max")``)
    stream          raw handle of the stream the collectives are issued on; 0 = torch's current stream (run under
                    ``with torch.cuda.stream(side)``: the legacy default stream is refused, see ``_stream_for``).
    """
    stream = _stream_for(stream, bins)
    with _torch_on(stream, bins):
        return _sharded_group_by(engine, query, group_column, bins, all_reduce_sum, all_reduce_max, stream)


def _sharded_group_by(engine, query, group_column, bins, all_reduce_sum, all_reduce_max, stream):
    lo, hi = engine.group_key_range(group_column)
    rng = bins.new_tensor([-float(lo), float(hi)])
    all_reduce_max(rng)
    kmin, kmax = -int(rng[0].item()), int(rng[1].item())
    if kmax < kmin:
        return []  # an empty table
    nbins = kmax - kmin + 1
    if bins.numel() < 4 * nbins:
        raise ValueError(f"bin buffer holds {bins.numel()} doubles, {4 * nbins} needed")
    b = bins[: 4 * nbins]
    engine.grouped_enqueue_bins(query, group_column, kmin, nbins, b.data_ptr(), stream)
    all_reduce_sum(b)
    return engine.grouped_finish(query, kmin, nbins, b.data_ptr(), stream)


# ---- the variance-aware samplers over a sharded table (SURVEY 8e "what does not shard") ---------------------------------
# Both need one fact about the WHOLE table before a shard can plan (include/aqe_hip.h, the block above aqe_zone_moments).
# The exchanges below are small host arrays, once per table and query shape — `host_all_reduce_sum(a) -> a summed over the
# ranks` for a float64 numpy array (torch_host_all_reduce) — the plan they end in runs like any other (ShardedQuery).

def sharded_adaptive_plan(engine, query, host_all_reduce_sum: Callable):
    """adaptive_block_sample (custom_bplus_db.cpp:1273-1329) over a sharded table: the ten zones' raw moments are additive,
    so ONE all-reduce of 30 doubles gives every rank the reference's zone variances (var = Q/n - (S/n)^2, DB.cpp:1291-1308);
    every rank then plans the same blocks and keeps the part inside its rows."""
    import numpy as np
    m = np.asarray(host_all_reduce_sum(np.ascontiguousarray(engine.zone_moments(), dtype=np.float64).reshape(-1))).reshape(10, 3)
    cnt = m[:, 0]
    mean = m[:, 1] / cnt
    engine.set_zone_variances(m[:, 2] / cnt - mean * mean)
    return engine.plan(query)


_KEY_TOP = 1 << 63


def _key_to_double(keys):
    """Inverse of the order-preserving map double -> uint64 (sign bit flipped for positives, all bits for negatives)."""
    import numpy as np
    k = np.asarray(keys, dtype=np.uint64)
    pos = (k & np.uint64(_KEY_TOP)) != 0
    bits = np.where(pos, k ^ np.uint64(_KEY_TOP), ~k)
    return bits.view(np.float64)


def _double_to_key(values):
    import numpy as np
    b = np.asarray(values, dtype=np.float64).view(np.uint64)
    neg = (b & np.uint64(_KEY_TOP)) != 0
    return np.where(neg, ~b, b | np.uint64(_KEY_TOP))


def sorted_positions_to_local(engine, positions, n_global: int, host_all_reduce_sum: Callable, rank: int, world: int):
    """Where positions of the GLOBAL amount-sorted order fall in this rank's own sorted column: L(p) = how many of the first
    p rows of the global order this rank holds (so the global run [s, e) is the local run [L(s), L(e)), and the local runs of
    all ranks add up to it).  The value at global position p is found by bisection over the doubles' bit patterns — 64 steps,
    each one `engine.sorted_counts` and one all-reduce of the counts, for all positions at once — and rows that tie with it
    are dealt to the ranks in rank order (tied rows hold the same amount: any of them gives the same aggregate)."""
    import numpy as np
    pos = np.asarray(positions, dtype=np.uint64)
    M = len(pos)
    if M == 0:
        return np.zeros(0, dtype=np.uint64)
    inside = np.minimum(pos, np.uint64(max(n_global, 1) - 1))  # (position N itself is past the last row: fixed below)
    want = inside.astype(np.float64) + 1.0
    lo = np.full(M, _double_to_key(np.array([-np.inf]))[0], dtype=np.uint64)
    hi = np.full(M, _double_to_key(np.array([np.inf]))[0], dtype=np.uint64)
    for _ in range(64):  # the smallest value v with (rows <= v over all ranks) >= p + 1 is the value at position p
        mid = lo + (hi - lo) // np.uint64(2)
        _, le = engine.sorted_counts(_key_to_double(mid))
        ok = np.asarray(host_all_reduce_sum(le.astype(np.float64))) >= want
        hi = np.where(ok, mid, hi)
        lo = np.where(ok, lo, mid + np.uint64(1))
    lt, le = engine.sorted_counts(_key_to_double(lo))
    pack = np.zeros((world + 1, M), dtype=np.float64)
    pack[rank] = (le - lt).astype(np.float64)  # this rank's rows that tie with the value
    pack[world] = lt.astype(np.float64)
    pack = np.asarray(host_all_reduce_sum(pack.reshape(-1))).reshape(world + 1, M)
    need = inside.astype(np.float64) - pack[world]          # tied rows that come before position p, over all ranks
    before = pack[:rank].sum(axis=0)                         # ... of which the lower ranks supply these
    mine = np.clip(need - before, 0.0, pack[rank])
    local = lt + mine.astype(np.uint64)
    n_local = int(engine.info().local_rows)
    return np.where(pos >= np.uint64(n_global), np.uint64(n_local), local).astype(np.uint64)


def _family_runs(fams):
    """(start, length) arrays of the maximal runs of consecutive rows of step-1 families, in family order."""
    import numpy as np
    starts, lens = [], []
    for f in fams:
        if f.ord_hi <= f.ord_lo:
            continue
        if f.step != 1:
            raise ValueError("runs of consecutive rows only")
        if f.seg_len >= f.ord_hi:  # one segment (seg_len may be 2^64 - 1: no products with it)
            starts.append(np.array([f.row0 + f.ord_lo], dtype=np.uint64))
            lens.append(np.array([f.ord_hi - f.ord_lo], dtype=np.uint64))
            continue
        j0, j1 = f.ord_lo // f.seg_len, (f.ord_hi - 1) // f.seg_len
        j = np.arange(j0, j1 + 1, dtype=np.uint64)
        o0 = np.maximum(j * np.uint64(f.seg_len), np.uint64(f.ord_lo))
        o1 = np.minimum((j + np.uint64(1)) * np.uint64(f.seg_len), np.uint64(f.ord_hi))
        starts.append(np.uint64(f.row0) + j * np.uint64(f.pitch) + (o0 - j * np.uint64(f.seg_len)))
        lens.append(o1 - o0)
    if not starts:
        return np.zeros(0, dtype=np.uint64), np.zeros(0, dtype=np.uint64)
    return np.concatenate(starts), np.concatenate(lens)


def sharded_stratified_plan(engine, query, host_all_reduce_sum: Callable, rank: int, world: int):
    """stratified_block_sample (custom_bplus_db.cpp:1331-1379) over a sharded table.  The reference sorts a copy of every
    record by amount (DB.cpp:1342-1345) and takes blocks of that order; here every rank sorts its own rows, the blocks are
    planned in GLOBAL sorted positions (host planner, no data needed), and `sorted_positions_to_local` turns each block
    into the run of this rank's sorted column that lies inside it.  The rows taken over all ranks are exactly the rows of
    the global blocks (up to which of several rows with the same amount is taken)."""
    from . import _native as nat
    n_global = int(engine.info().global_rows)
    fams, _, samples = nat.plan_families(query, n_global)
    starts, lens = _family_runs(fams)
    import numpy as np
    bounds = np.unique(np.concatenate([starts, starts + lens])) if len(starts) else np.zeros(0, dtype=np.uint64)
    local = sorted_positions_to_local(engine, bounds, n_global, host_all_reduce_sum, rank, world)
    a = local[np.searchsorted(bounds, starts)] if len(starts) else local
    b = local[np.searchsorted(bounds, starts + lens)] if len(starts) else local
    runs = []
    for x, y in zip(a.tolist(), b.tolist()):
        if y > x:
            runs.append(nat.Family(row0=x, pitch=0, seg_len=y - x, step=1, ord_lo=0, ord_hi=y - x))
    return engine.plan_families(query, runs, samples, on_sorted=True)


def torch_host_all_reduce(group=None, device=None) -> Callable:
    """host_all_reduce_sum for the planners above through torch.distributed: float64 numpy array in, its sum over the ranks
    out (staged through `device` — the rank's GPU for backend "nccl", which only moves device tensors)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    def _ar(a):
        t = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64).copy())
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        return t.cpu().numpy()

    return _ar


def torch_all_reduce(group=None, op: str = "sum") -> Callable:
    """In-place all-reduce (SUM, or MAX with op="max") through torch.distributed (backend "nccl" is RCCL on ROCm;
    "gloo" on CPU)."""
    import torch.distributed as dist
    red = {"sum": dist.ReduceOp.SUM, "max": dist.ReduceOp.MAX}[op]

    def _ar(t):
        dist.all_reduce(t, op=red, group=group)

    return _ar


def native_all_reduce(comm, stream: int, op: str = "sum") -> Callable:
    """In-place all-reduce through the library's own RCCL communicator (engine.Comm, aqe_comm_*): what a C++ host of
    the C ABI uses; no torch.distributed involved in the data path."""
    fn = comm.all_reduce_sum if op == "sum" else comm.all_reduce_max

    def _ar(t):
        fn(t.data_ptr(), t.numel(), stream)

    return _ar


def comm_from_torch_group(engine, group=None):
    """An engine.Comm over the ranks of a torch.distributed group: rank 0 draws the RCCL unique id, the group
    broadcasts its 128 bytes (any backend — this is the out-of-band exchange), every rank joins."""
    import torch.distributed as dist
    from .engine import Comm
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    ident = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(ident, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    return Comm(engine, ident[0], world, rank)


def mailbox_from_torch_group(engine, group=None):
    """An engine.Mailbox over the ranks of a torch.distributed group (one process per GPU): the 64-byte IPC handles travel
    through the group once (any backend), every rank maps every peer's mailbox."""
    import torch.distributed as dist
    from .engine import Mailbox
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mb = Mailbox(engine, world, rank)
    handles = [None] * world
    dist.all_gather_object(handles, mb.handle(), group=group)
    mb.connect(handles)
    dist.barrier(group=group)  # every rank has mapped its peers before the first store
    return mb


def mailbox_all_reduce(mailbox, stream: int) -> Callable:
    """In-place all-reduce SUM through the peer-mapped mailbox (engine.Mailbox): one single-workgroup launch per rank on
    `stream`, no library collective.  For tensors of at most 4096 doubles — the moment vectors and round totals of this path."""

    def _ar(t):
        mailbox.all_reduce_sum(t.data_ptr(), t.numel(), stream)

    return _ar
