"""The N>1 path on CPU: world_size-2 (and 3) `gloo` process groups run the real ShardedQuery orchestration
(one all-reduce of the 8-double moment vector per convergence step) over oracle-backed shard plans, and
must reproduce the single-process oracle answer on every rank."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, specs, out_dir, batched=False):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_engine import OracleShardPlan
    from approximatequeryengine_amd.distributed import ShardedQuery, shard_bounds
    from oracle.pyoracle import Oracle
    o = Oracle()
    lo, hi = shard_bounds(n, world, rank)
    rows = o.synth(hi - lo, 42, first=lo)       # each rank generates only its own shard
    head = o.synth(min(n, 1024), 42)
    shift = float(head["amount"].mean())          # every shard of a table must use the same shift
    results = []
    for spec in specs:
        calls = [0]

        def all_reduce(t):
            calls[0] += 1
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

        plan = OracleShardPlan(o, rows, lo, n, shift, spec)
        vec = torch.zeros(max(8, plan.totals_len), dtype=torch.float64)
        sq = ShardedQuery(plan, vec, all_reduce, batched=batched)
        res = sq.run()
        res["collectives"] = calls[0]
        res["batched"] = sq.batched
        res["steps"] = plan.rounds + (1 if plan.has_topup else 0)
        results.append(res)
    torch.save(results, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


SPECS = [
    ("stride", 1.0),
    ("clt", 20.0, 0.95, 10, 4, 1.0, 256, 2),     # converges early -> top-up
    ("clt", 20.0, 0.95, 10, 4, 0.0, 4096, 4),    # never converges
    ("clt", 10.0, 0.95, 10, 6, 0.5, 64, 2),
    ("clt", 20.0, 0.95, 10, 4, 5.0, 64, 2),      # stops after ~768 rows < base/4 -> the top-up is due
]


@pytest.mark.parametrize("world,batched", [(2, False), (3, False), (2, True), (4, False), (4, True)])
def test_sharded_query_over_gloo_matches_single_process_oracle(oracle, table, tmp_path, world, batched):
    n = 200_003
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, SPECS, str(tmp_path), batched), nprocs=world, join=True)
    per_rank = [torch.load(tmp_path / f"r{r}.pt", weights_only=False) for r in range(world)]
    rows = table(n)
    for i, spec in enumerate(SPECS):
        got = [pr[i] for pr in per_rank]
        for g in got[1:]:  # every rank folds the same reduced vector -> identical answers
            assert g == got[0]
        g = got[0]
        if not g["batched"]:
            assert g["collectives"] == g["steps"]  # one all-reduce per convergence step (+ the top-up step)
        if spec[0] == "stride":
            idx = oracle.idx_memory_stride(n, spec[1])
            m = oracle.moments_idx(rows, idx)
            assert g["n"] == m.n and abs(g["sum"] - m.sum) <= 1e-12 * abs(m.sum)
        else:
            _, pct, conf, ci, T, e, R0, growth = spec
            rc, want, _ = oracle.clt_run(rows, pct, conf, ci, T, e, R0=R0, growth=growth)
            assert rc == 0
            assert (g["n"], g["converged"], g["rounds"], g["topup"]) == (want.final.n, want.converged, want.rounds, want.topup)
            if g["batched"]:  # one all-reduce per query; one more only when the top-up was due
                assert g["collectives"] == 1 + (1 if want.topup else 0) and g["topup_pending"] == 0
            assert abs(g["sum"] - want.final.sum) <= 1e-12 * abs(want.final.sum)
            assert abs(g["m2"] - want.final.m2) <= 1e-9 * abs(want.final.m2)


def _batch_worker(rank, world, port, n, specs, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_engine import OracleShardPlan
    from approximatequeryengine_amd.distributed import PipelinedBatches, ShardedBatch, shard_bounds
    from oracle.pyoracle import Oracle
    o = Oracle()
    lo, hi = shard_bounds(n, world, rank)
    rows = o.synth(hi - lo, 42, first=lo)
    shift = float(o.synth(min(n, 1024), 42)["amount"].mean())
    calls = [0]

    def all_reduce(t):
        calls[0] += 1
        dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def make():
        plans = [OracleShardPlan(o, rows, lo, n, shift, spec) for spec in specs]
        buf = torch.zeros(len(plans), max(p.totals_len for p in plans) + 8, dtype=torch.float64)
        return ShardedBatch(plans, buf, all_reduce)

    # (a) one batch, one collective for all of its queries; run() finishes the due top-ups with ONE more collective
    sb = make()
    sb.enqueue()
    marks = [r["topup_pending"] for r in sb.fetch()]
    c_step = calls[0]
    out = sb.run()
    c_run = calls[0] - c_step
    # (b) two batches software-pipelined over several steps: a step's collective is issued after the NEXT step's sweeps
    pipe = PipelinedBatches([make(), make()])
    before = calls[0]
    for _ in range(5):
        pipe.enqueue()
    piped = pipe.fetch()
    torch.save({"marks": marks, "c_step": c_step, "c_run": c_run, "out": out, "piped": piped, "c_pipe": calls[0] - before},
               os.path.join(out_dir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_batch_and_pipeline_over_gloo(oracle, table, tmp_path, world):
    """ShardedBatch (one all-reduce per batch of queries) and PipelinedBatches over gloo: every rank gets the
    single-process oracle's answer for every query; a due top-up (DB.cpp:1031-1040) shows as the same mark on every
    rank and is finished by run() with one more collective for the whole batch."""
    n = 200_003
    specs = [s_ for s_ in SPECS if s_[0] == "clt"]
    port = _free_port()
    mp.spawn(_batch_worker, args=(world, port, n, specs, str(tmp_path)), nprocs=world, join=True)
    per_rank = [torch.load(tmp_path / f"b{r}.pt", weights_only=False) for r in range(world)]
    rows = table(n)
    wants = []
    for _, pct, conf, ci, T, e, R0, growth in specs:
        rc, want, _ = oracle.clt_run(rows, pct, conf, ci, T, e, R0=R0, growth=growth)
        assert rc == 0
        wants.append(want)
    for pr in per_rank:
        assert pr == per_rank[0]  # every rank folds the same reduced buffer
    g = per_rank[0]
    assert g["marks"] == [1 if w.topup else 0 for w in wants] and any(g["marks"]) and not all(g["marks"])
    assert g["c_step"] == 1                      # ONE collective for the whole batch
    assert g["c_run"] == 2                       # run(): the batch's collective + one more for all due top-ups
    assert g["c_pipe"] == 5 and len(g["piped"]) == 2 * len(specs)
    for got, w in zip(g["out"], wants):
        assert (got["n"], got["converged"], got["rounds"], got["topup"], got["topup_pending"]) == (w.final.n, w.converged, w.rounds, w.topup, 0)
        assert abs(got["sum"] - w.final.sum) <= 1e-12 * abs(w.final.sum)
    for got, w in zip(g["piped"], wants + wants):
        assert (got["converged"], got["rounds"], got["topup_pending"]) == (w.converged, w.rounds, 1 if w.topup else 0)
        if not w.topup:
            assert got["n"] == w.final.n and abs(got["sum"] - w.final.sum) <= 1e-12 * abs(w.final.sum)


def test_shard_bounds_are_a_proper_partition():
    from approximatequeryengine_amd.distributed import shard_bounds
    for n in (0, 1, 7, 100_007, 10_000_000):
        for g in (1, 2, 3, 8):
            b = [shard_bounds(n, g, r) for r in range(g)]
            assert b[0][0] == 0 and b[-1][1] == n and all(b[i][1] == b[i + 1][0] for i in range(g - 1))
            assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


# ---- the variance-aware samplers over a sharded table: zone variances and sorted positions are agreed over the group ----
from helpers import VARIANCE_AWARE, va_table as _va_table


def _va_worker(rank, world, port, n, ties, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from fake_engine import OracleShardEngine
    from helpers import VARIANCE_AWARE, va_table as _va_table
    from approximatequeryengine_amd import _native as nat
    from approximatequeryengine_amd.distributed import (ShardedQuery, shard_bounds, sharded_adaptive_plan, sharded_stratified_plan,
                                                        torch_host_all_reduce)
    from approximatequeryengine_amd.engine import make_query
    from oracle.pyoracle import Oracle
    o = Oracle()
    lo, hi = shard_bounds(n, world, rank)
    eng = OracleShardEngine(_va_table(o, n, lo, hi, ties), lo, n, 500.5)
    host_ar = torch_host_all_reduce()
    results = []
    for kind, pct, a, b in VARIANCE_AWARE:
        if kind == "adaptive":
            q = make_query(nat.M_ADAPTIVE_BLOCK, pct, block_size=a, block_size_max=b)
            plan = sharded_adaptive_plan(eng, q, host_ar)
        else:
            q = make_query(nat.M_STRATIFIED_BLOCK, pct, block_size=a, num_threads=b)
            plan = sharded_stratified_plan(eng, q, host_ar, rank, world)
        vec = torch.zeros(8, dtype=torch.float64)
        res = ShardedQuery(plan, vec, lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)).run()
        res["local_rows"] = len(plan.amounts)
        results.append(res)
    torch.save(results, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ties", [(2, False), (3, True), (4, False), (2, True)])
def test_variance_aware_samplers_over_gloo_match_the_whole_table(oracle, tmp_path, world, ties):
    """adaptive_block_sample and stratified_block_sample on a table sharded over `world` ranks (distributed.py:
    sharded_adaptive_plan, sharded_stratified_plan) take exactly the rows the oracle takes on the whole table — the same
    count, the same sum — whatever the shard boundaries, also when most amounts tie."""
    n = 120_007
    port = _free_port()
    mp.spawn(_va_worker, args=(world, port, n, ties, str(tmp_path)), nprocs=world, join=True)
    per_rank = [torch.load(tmp_path / f"r{r}.pt", weights_only=False) for r in range(world)]
    rows = _va_table(oracle, n, 0, n, ties)
    for i, (kind, pct, a, b) in enumerate(VARIANCE_AWARE):
        got = [pr[i] for pr in per_rank]
        for g in got[1:]:
            assert {k: v for k, v in g.items() if k != "local_rows"} == {k: v for k, v in got[0].items() if k != "local_rows"}
        idx = oracle.idx_adaptive_block(rows, pct, a, b) if kind == "adaptive" else oracle.idx_stratified_block(rows, pct, a, b)
        m = oracle.moments_idx(rows, idx)
        g = got[0]
        assert m.n > 0 and g["n"] == m.n, (kind, pct, a, b, g["n"], m.n)
        assert sum(x["local_rows"] for x in got) == m.n
        assert abs(g["sum"] - m.sum) <= 1e-12 * abs(m.sum), (kind, pct, a, b)
        assert abs(g["m2"] - m.m2) <= 1e-9 * abs(m.m2)


def test_sorted_position_helpers_on_the_host():
    """The pieces of sharded_stratified_plan that are plain arithmetic: the order-preserving map double <-> uint64 the
    bisection walks, the runs of consecutive rows of a family, and the global -> local position map on one process with three
    in-memory shards (every collective a sum over the shards), ties included."""
    from fake_engine import OracleShardEngine
    from approximatequeryengine_amd import _native as nat
    from approximatequeryengine_amd.distributed import _double_to_key, _family_runs, _key_to_double, sorted_positions_to_local
    v = np.array([-np.inf, -1e300, -2.5, -5e-324, -0.0, 0.0, 5e-324, 1.0, 1.0000000000000002, 1e300, np.inf])
    k = _double_to_key(v)
    assert np.all(k[1:] >= k[:-1]) and np.array_equal(_key_to_double(k).view(np.uint64), v.view(np.uint64))
    fams = [nat.Family(row0=5, pitch=0, seg_len=2**64 - 1, step=1, ord_lo=3, ord_hi=10), nat.Family(row0=100, pitch=50, seg_len=7, step=1, ord_lo=5, ord_hi=16)]
    s, l = _family_runs(fams)
    assert s.tolist() == [8, 105, 150, 200] and l.tolist() == [7, 2, 7, 2]
    with pytest.raises(ValueError):
        _family_runs([nat.Family(row0=0, pitch=0, seg_len=8, step=2, ord_lo=0, ord_hi=8)])
    # three shards of a table whose amounts tie by the dozen; the "group" is a loop over the shards
    rng = np.random.default_rng(3)
    n = 5_000
    amount = np.round(rng.normal(500.0, 40.0, n))
    cuts = [0, 1_700, 1_701, n]
    rows = np.zeros(n, dtype=[("amount", "f8")])
    rows["amount"] = amount
    engs = [OracleShardEngine(rows[a:b], a, n, 500.0) for a, b in zip(cuts[:-1], cuts[1:])]
    pos = np.unique(np.concatenate([rng.integers(0, n + 1, 200), [0, 1, n - 1, n]])).astype(np.uint64)
    # run the three "ranks" in lockstep: every host_all_reduce call sums what the three contribute at the same step
    import threading
    box, bar = [None] * 3, threading.Barrier(3)
    out = [None] * 3

    def rank(r):
        def ar(a):
            box[r] = np.asarray(a, dtype=np.float64).copy()
            bar.wait()
            tot = box[0] + box[1] + box[2]
            bar.wait()
            return tot
        out[r] = sorted_positions_to_local(engs[r], pos, n, ar, r, 3)

    ts = [threading.Thread(target=rank, args=(r,)) for r in range(3)]
    [t.start() for t in ts]
    [t.join(timeout=120) for t in ts]
    tot = out[0].astype(np.int64) + out[1].astype(np.int64) + out[2].astype(np.int64)
    assert np.array_equal(tot, pos.astype(np.int64))  # the local positions add up to the global one
    srt = np.sort(amount)
    for r, e in enumerate(engs):
        L = out[r].astype(np.int64)
        assert np.all(L[1:] >= L[:-1]) and L[0] == 0 and L[-1] == len(e.sorted)
        # the first L rows of the shard's sorted column are rows of the global prefix: none exceeds the value at the global position
        for p, l in zip(pos.astype(np.int64), L):
            if 0 < l and p < n:
                assert e.sorted[l - 1] <= srt[p]
            if l < len(e.sorted) and p > 0:
                assert e.sorted[l] >= srt[p - 1]
