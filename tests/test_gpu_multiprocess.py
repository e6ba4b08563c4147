"""The N > 1 path with the REAL engines: 2 and 4 freshly spawned processes share cuda:0, each stages only its
`shard_bounds` region through the C ABI (libaqe_hip.so), sweeps it with the HIP kernels, and `gloo` carries the moment
vectors between the processes (RCCL refuses two ranks on one GPU; on a multi-GPU node the same code runs over backend
"nccl").  Every rank must get the answer of a single engine holding the whole table — and of the oracle.

Replaces, across processes: the reference's in-process merges (custom_bplus_db.cpp:948-951, 966-967, 2031-2036) over its
region partition (custom_bplus_db.cpp:1903-1921)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from helpers import VARIANCE_AWARE, rel, va_table

N = 400_003
N_VA = 300_007  # the table of the variance-aware samplers (helpers.va_table: zones of different spread, tied amounts)
CLT_SPECS = [  # (pct, T, e, R0, growth)
    (20.0, 4, 1.0, 256, 2),     # converges early -> the top-up is due
    (20.0, 4, 0.0, 4096, 4),    # never converges
    (10.0, 6, 0.5, 64, 2),
    (20.0, 8, 5.0, 64, 2),      # stops after a few hundred rows: top-up
]


FACADE_CALLS = [  # (aggregate, keywords of approx()): what one engine holding the whole table must answer the same way
    ("SUM", dict(method="stride", sample_percent=1.0)),
    ("AVG", dict(method="random", sample_percent=2.0, seed=9)),
    ("SUM", dict(method="random_device", sample_percent=1.0, seed=5)),
    ("SUM", dict(method="block", sample_percent=5.0, where=(250.0, 750.0), convention="cpp")),
    ("COUNT", dict(method="parallel_block", sample_percent=3.0, num_threads=6)),
    ("AVG", dict(method="region", sample_percent=2.0, seed=11)),
    ("AVG", dict(method="clt", error_percent=1.0)),
    ("SUM", dict(method="clt", error_percent=0.01, num_threads=6)),
    ("AVG", dict(method="exact", where=(100.0, 900.0))),
    ("SUM", dict(method="adaptive_block", sample_percent=5.0)),
    ("AVG", dict(method="stratified_block", sample_percent=2.0, block_size=100, num_threads=7)),
    ("SUM", dict(method="stride", sample_percent=5.0, id_between=(90_001, 250_000))),   # key bounds -> a row window that spans shards
    ("AVG", dict(method="clt", error_percent=2.0, id_between=(10, 299_000))),
    ("SUM", dict(method="exact", id_between=(120_000, 120_500))),                      # ... and one inside a single shard
]
FACADE_BATCH = [dict(agg="AVG", method="clt", error_percent=1.0), dict(agg="SUM", method="clt", error_percent=0.01, num_threads=8),
                dict(agg="SUM", method="block", sample_percent=1.0, where=(250.0, 750.0)), dict(agg="AVG", method="clt", error_percent=5.0)]
FACADE_CLI = [["SELECT SUM(amount) FROM sales", "--s", "1", "--ci", "--compare"], ["SELECT AVG(amount) FROM sales", "--e", "2"],
              ["SELECT COUNT(*) FROM sales"], ["SELECT AVG(amount) FROM sales GROUP BY region", "--s", "10", "--ci"]]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _queries(nat, make_query):
    qs = [make_query(nat.M_MEMORY_STRIDE, 1.0), make_query(nat.M_BLOCK, 5.0, where=(250.0, 750.0), convention=nat.EST_CPP),
          make_query(nat.M_RANDOM_POINTER, 2.0, seed=9), make_query(nat.M_EXACT, 100.0, agg=nat.AVG)]
    qs += [make_query(nat.M_CLT_DUAL_POINTER, pct, agg=nat.AVG, num_threads=T, max_error_percent=e, clt_round0=r0, clt_growth=g)
           for pct, T, e, r0, g in CLT_SPECS]
    return qs


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from approximatequeryengine_amd import _native as nat
    from approximatequeryengine_amd.distributed import PipelinedBatches, ShardedBatch, ShardedQuery, shard_bounds, sharded_group_by, torch_all_reduce
    from approximatequeryengine_amd.engine import Batch, Engine, make_query
    torch.cuda.set_device(0)
    lo, hi = shard_bounds(N, world, rank)
    calls = [0]
    ar_sum, ar_max = torch_all_reduce(), torch_all_reduce(op="max")

    def all_reduce(t):
        calls[0] += 1
        ar_sum(t)

    out = {"single": [], "batched": [], "kernels": []}
    side = torch.cuda.Stream()
    with Engine(0) as eng, torch.cuda.stream(side):
        eng.generate_synthetic(hi - lo, shard_lo=lo, n_global=N, seed=42, keep_aos=True)  # each rank holds only its region
        st = side.cuda_stream
        qs = _queries(nat, make_query)
        # (a) ShardedQuery: one collective per convergence step, then the batched form (ONE collective per query)
        for q in qs:
            plan = eng.plan(q)
            vec = torch.zeros(max(nat.MOMENT_VEC, plan.totals_len), dtype=torch.float64, device="cuda")
            for batched in (False, True):
                if batched and not plan.totals_len:
                    out["batched"].append(None)
                    continue
                before = calls[0]
                sq = ShardedQuery(plan, vec, all_reduce, stream=st, batched=batched)
                r = sq.run().as_dict()
                r["collectives"] = calls[0] - before
                r["steps"] = plan.rounds + (1 if plan.has_topup else 0)
                out["batched" if batched else "single"].append(r)
            out["kernels"].append(plan.last_kernel())
            plan.close()
        # (b) ShardedBatch over the CLT queries: the sweeps of the whole batch are ONE launch, ONE collective, ONE replay
        #     launch; run() finishes the due top-ups with one more collective for all of them
        clt = [q for q in qs if q.method == nat.M_CLT_DUAL_POINTER]

        def make():
            plans = [eng.plan(q) for q in clt]
            buf = torch.zeros(len(plans), max(p.totals_len for p in plans), dtype=torch.float64, device="cuda")
            return ShardedBatch(plans, buf, all_reduce, stream=st, batch=Batch(plans))

        sb = make()
        sb.enqueue()
        out["marks"] = [int(r.topup_pending) for r in sb.fetch()]
        before = calls[0]
        out["batch"] = [r.as_dict() for r in sb.run()]
        out["c_run"] = calls[0] - before
        # (c) two batches software-pipelined over five steps (a step's collective is issued after the NEXT step's sweeps)
        pipe = PipelinedBatches([make(), make()])
        before = calls[0]
        for _ in range(5):
            pipe.enqueue()
        out["piped"] = [r.as_dict() for r in pipe.fetch()]
        out["c_pipe"] = calls[0] - before
        # (d) GROUP BY across the ranks: key range agreed (MAX), bins all-reduced (SUM), every rank finishes
        bins = torch.zeros(4 * 1024, dtype=torch.float64, device="cuda")
        gq = make_query(nat.M_ROWID_MOD, 10.0, agg=nat.AVG)
        out["groups"] = [g.as_dict() for g in sharded_group_by(eng, gq, nat.GROUP_PRODUCT, bins, ar_sum, ar_max, stream=st)]
        torch.cuda.synchronize()
    # (e) adaptive_block_sample / stratified_block_sample over a sharded table: the zone variances and the global sorted
    #     positions are agreed over the group first (distributed.py), then the plan runs like any other
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from approximatequeryengine_amd.distributed import sharded_adaptive_plan, sharded_stratified_plan, torch_host_all_reduce
    from oracle.pyoracle import Oracle  # (test data only: the table's rows)
    lo2, hi2 = shard_bounds(N_VA, world, rank)
    host_ar = torch_host_all_reduce()
    out["va"] = []
    with Engine(0) as eng, torch.cuda.stream(side):
        eng.stage_records(va_table(Oracle(), N_VA, lo2, hi2, ties=True), shard_lo=lo2, n_global=N_VA)
        eng.set_shift(500.5)  # every shard of one table uses the same shift
        vec = torch.zeros(nat.MOMENT_VEC, dtype=torch.float64, device="cuda")
        for kind, pct, a, b in VARIANCE_AWARE:
            if kind == "adaptive":
                plan = sharded_adaptive_plan(eng, make_query(nat.M_ADAPTIVE_BLOCK, pct, block_size=a, block_size_max=b), host_ar)
            else:
                plan = sharded_stratified_plan(eng, make_query(nat.M_STRATIFIED_BLOCK, pct, block_size=a, num_threads=b), host_ar, rank, world)
            r = ShardedQuery(plan, vec, all_reduce, stream=side.cuda_stream).run().as_dict()
            out["va"].append(r)
            plan.close()
        # (f) the peer-mapped mailbox instead of the library collective (the IPC handles travel through the group once): the
        #     same sharded queries, stepwise and batched, and a batch per collective
        from approximatequeryengine_amd.distributed import mailbox_all_reduce, mailbox_from_torch_group
        mb = mailbox_from_torch_group(eng)
        mar = mailbox_all_reduce(mb, side.cuda_stream)
        out["mailbox"] = []
        for pct, T, e, r0, g in CLT_SPECS[:3]:
            plan = eng.plan(make_query(nat.M_CLT_DUAL_POINTER, pct, agg=nat.AVG, num_threads=T, max_error_percent=e, clt_round0=r0, clt_growth=g))
            v2 = torch.zeros(max(nat.MOMENT_VEC, plan.totals_len), dtype=torch.float64, device="cuda")
            for batched in (False, True):
                a = ShardedQuery(plan, v2, mar, stream=side.cuda_stream, batched=batched).run().as_dict()
                b = ShardedQuery(plan, v2, all_reduce, stream=side.cuda_stream, batched=batched).run().as_dict()
                out["mailbox"].append((a, b))
            plan.close()
        # ... and the host side in C: aqe_plan_run_sharded over a communicator that wraps the mailbox
        from approximatequeryengine_amd.engine import Comm
        mc = Comm.over_mailbox(eng, mb)
        pct, T, e, r0, g = CLT_SPECS[0]
        plan = eng.plan(make_query(nat.M_CLT_DUAL_POINTER, pct, agg=nat.AVG, num_threads=T, max_error_percent=e, clt_round0=r0, clt_growth=g))
        v2 = torch.zeros(max(nat.MOMENT_VEC, plan.totals_len), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        out["mailbox_c"] = (mc.run_plan(plan, v2.data_ptr(), side.cuda_stream).as_dict(), out["mailbox"][1][0])
        plan.close()
        mc.close()
        out["mailbox_late"] = mb.late_ranks()
        dist.barrier()  # every rank is done with its peers' mailboxes before any is destroyed
        mb.close()
        # what a shard refuses by itself: the stratified sampler without the exchange
        try:
            eng.plan(make_query(nat.M_STRATIFIED_BLOCK, 2.0, block_size=100, num_threads=7))
            out["va_refused"] = False
        except nat.AqeError:
            out["va_refused"] = True
        torch.cuda.synchronize()
    # (g) the CustomBPlusDB query API over the sharded table (sharded_backend.ShardedBPlusDB): every rank opens the same file
    #     and stages only its region; the same calls on every rank, the same answers; then the CLI the same way
    import io
    from approximatequeryengine_amd import cli
    from approximatequeryengine_amd.sharded_backend import ShardedBPlusDB
    path = os.path.join(out_dir, "va.db")
    pick = lambda r: (r.value, r.ci_lower, r.ci_upper, int(r.n), int(r.converged), int(r.rounds), int(r.topup))
    for name, coll in (("db", "torch"), ("db_mailbox", "mailbox")):
        db = ShardedBPlusDB(device_id=0, collective=coll)
        assert db.open_database(path) and db.get_total_records() == N_VA and db.shard() == shard_bounds(N_VA, world, rank)
        o = {"exact": (db.sum_amount(), db.avg_amount(), db.count_records(), db.sum_amount_where(250.0, 750.0))}
        o["approx"] = [pick(db.approx(agg, **kw)) for agg, kw in FACADE_CALLS]
        o["batch"] = [pick(r) for r in db.approx_batch(FACADE_BATCH)]
        o["groups"] = {k: (g.value, g.ci_lower, g.ci_upper, g.n) for k, g in db.approx_group_by("AVG", "product_id", 10.0).items()}
        o["psum"] = db.parallel_sum_sample(5.0, seed=3)
        o["psum_unseeded"] = db.parallel_sum_sample(5.0)  # rank 0's seed for everybody
        from approximatequeryengine_amd.aqe_backend import CustomApproximateScheduler
        sch = CustomApproximateScheduler(db=db, seed=77)  # the scheduler façade over the sharded table
        o["sched"] = [(r.value, r.status.name, r.samples_used) for r in (sch.execute_sum_query("SELECT SUM(amount) FROM sales WHERE amount BETWEEN 100 AND 800", 5.0),
                                                                            sch.execute_avg_query("SELECT AVG(amount) FROM sales", 2.0), sch.execute_exact_sum())]
        for bad in (lambda: db.memory_stride_sample(1.0), lambda: db.insert_record(None)):
            try:
                bad()
                o["refused"] = False
            except NotImplementedError:
                o.setdefault("refused", True)
        db.close_database()
        out[name] = o
    texts = []
    for argv in FACADE_CLI:
        buf = io.StringIO()
        rc = cli.run(cli.build_parser().parse_args(argv + ["--db", path, "--backend", "gloo"]), buf)
        texts.append((rc, buf.getvalue()))
    out["cli"] = texts
    torch.save(out, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_real_engines_in_separate_processes_agree_with_one_engine_and_the_oracle(oracle, table, tmp_path, world):
    from approximatequeryengine_amd import _native as nat
    from approximatequeryengine_amd.engine import Engine, make_query
    assert oracle.file_write(tmp_path / "va.db", va_table(oracle, N_VA, 0, N_VA, ties=True)) == 0
    ctx = mp.get_context("spawn")  # children initialise the GPU themselves
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    per_rank = [torch.load(tmp_path / f"r{r}.pt", weights_only=False) for r in range(world)]
    strip = lambda d: {k: v for k, v in d.items() if k != "kernel_ms"} if isinstance(d, dict) else d
    for pr in per_rank[1:]:  # every rank folds the same reduced vectors: identical answers, bit for bit
        for key in ("single", "batched", "batch", "piped", "groups", "va"):
            assert [strip(x) for x in pr[key]] == [strip(x) for x in per_rank[0][key]], key
        assert pr["marks"] == per_rank[0]["marks"]
    g = per_rank[0]

    rows = table(N)
    qs = _queries(nat, make_query)
    with Engine(0) as whole:
        whole.generate_synthetic(N, seed=42, keep_aos=True)
        refs = [whole.reduce(q) for q in qs]
        gref = whole.reduce_grouped(make_query(nat.M_ROWID_MOD, 10.0, agg=nat.AVG), nat.GROUP_PRODUCT)

    def same(got, want):
        assert (got["n"], got["visited"], got["converged"], got["rounds"], got["topup"]) == (want.n, want.visited, want.converged, want.rounds, want.topup)
        assert rel(got["sum"], want.sum) <= 1e-12 and rel(got["value"], want.value) <= 1e-9
        assert rel(got["ci_lower"], want.ci_lower) <= 1e-9 and rel(got["ci_upper"], want.ci_upper) <= 1e-9

    for i, (q, want) in enumerate(zip(qs, refs)):
        s = g["single"][i]
        same(s, want)
        assert s["collectives"] == s["steps"]  # one all-reduce per convergence step (+ the top-up step)
        b = g["batched"][i]
        if b is not None:
            same(b, want)
            assert b["collectives"] == 1 + (1 if want.topup else 0) and b["topup_pending"] == 0
    # the oracle on the whole table (the single engine is itself held to it in test_gpu_parity.py)
    m = oracle.moments_idx(rows, oracle.idx_memory_stride(N, 1.0))
    assert g["single"][0]["n"] == m.n and rel(g["single"][0]["sum"], m.sum) <= 1e-12
    wants = []
    for pct, T, e, r0, gr in CLT_SPECS:
        rc, w, _ = oracle.clt_run(rows, pct, 0.95, 10, T, e, R0=r0, growth=gr)
        assert rc == 0
        wants.append(w)
    clt_got = [x for x, q in zip(g["single"], qs) if q.method == nat.M_CLT_DUAL_POINTER]
    for got, w in zip(clt_got, wants):
        assert (got["n"], got["converged"], got["rounds"], got["topup"]) == (w.final.n, w.converged, w.rounds, w.topup)
        assert rel(got["sum"], w.final.sum) <= 1e-12
    # ShardedBatch: marks, collectives, answers
    assert g["marks"] == [1 if w.topup else 0 for w in wants] and any(g["marks"]) and not all(g["marks"])
    assert g["c_run"] == 2 and g["c_pipe"] == 5 and len(g["piped"]) == 2 * len(CLT_SPECS)
    for got, w in zip(g["batch"], wants):
        assert (got["n"], got["converged"], got["rounds"], got["topup"], got["topup_pending"]) == (w.final.n, w.converged, w.rounds, w.topup, 0)
        assert rel(got["sum"], w.final.sum) <= 1e-12
    for got, w in zip(g["piped"], wants + wants):
        assert (got["converged"], got["rounds"], got["topup_pending"]) == (w.converged, w.rounds, 1 if w.topup else 0)
        if not w.topup:
            assert got["n"] == w.final.n and rel(got["sum"], w.final.sum) <= 1e-12
    # GROUP BY
    assert [(x["key"], x["n"], x["visited"]) for x in g["groups"]] == [(w.key, w.n, w.visited) for w in gref]
    for x, w in zip(g["groups"], gref):
        assert rel(x["sum"], w.sum) <= 1e-12 and rel(x["value"], w.value) <= 1e-9 and rel(x["ci_lower"], w.ci_lower) <= 1e-8
    # the peer-mapped mailbox gives what the library collective gives (the sum's order of additions may differ: 1e-12)
    assert all(pr["mailbox_late"] == 0 for pr in per_rank)
    for pr in per_rank:
        a, b = pr["mailbox_c"]  # (the C host path against distributed.ShardedQuery, batched form, same collective)
        assert strip(a) == strip(b), (a, b)
    for pr in per_rank:
        for (a, b), (a0, _) in zip(pr["mailbox"], per_rank[0]["mailbox"]):
            assert strip(a) == strip(a0)  # every rank adds the slots in rank order: identical, bit for bit
            assert (a["n"], a["visited"], a["converged"], a["rounds"], a["topup"]) == (b["n"], b["visited"], b["converged"], b["rounds"], b["topup"])
            assert rel(a["sum"], b["sum"]) <= 1e-12 and rel(a["value"], b["value"]) <= 1e-9 and rel(a["ci_lower"], b["ci_lower"]) <= 1e-9
    # the variance-aware samplers: one engine holding the whole table, and the oracle on the same rows
    assert g["va_refused"]
    va_rows = va_table(oracle, N_VA, 0, N_VA, ties=True)
    with Engine(0) as whole:
        whole.stage_records(va_rows)
        whole.set_shift(500.5)
        for got, (kind, pct, a, b) in zip(g["va"], VARIANCE_AWARE):
            if kind == "adaptive":
                want = whole.reduce(make_query(nat.M_ADAPTIVE_BLOCK, pct, block_size=a, block_size_max=b))
                idx = oracle.idx_adaptive_block(va_rows, pct, a, b)
            else:
                want = whole.reduce(make_query(nat.M_STRATIFIED_BLOCK, pct, block_size=a, num_threads=b))
                idx = oracle.idx_stratified_block(va_rows, pct, a, b)
            m = oracle.moments_idx(va_rows, idx)
            assert m.n > 0 and got["n"] == want.n == m.n, (kind, pct, a, b, got["n"], want.n, m.n)
            assert rel(got["sum"], m.sum) <= 1e-12 and rel(got["sum"], want.sum) <= 1e-12
            assert rel(got["value"], want.value) <= 1e-9 and rel(got["ci_lower"], want.ci_lower) <= 1e-8 and rel(got["ci_upper"], want.ci_upper) <= 1e-8
    # the façade over the sharded table (ShardedBPlusDB, with torch's collective and with the mailbox) and the CLI under a process group
    import io
    import re
    from approximatequeryengine_amd import aqe_backend, cli
    for name in ("db", "db_mailbox"):
        for pr in per_rank[1:]:
            assert pr[name] == per_rank[0][name], name  # every rank: the same answers, bit for bit
    one = aqe_backend.CustomBPlusDB()
    assert one.open_database(str(tmp_path / "va.db"))
    one._path = ""
    same_t = lambda a, b, tol: all((x == y) if isinstance(x, int) else rel(x, y) <= tol for x, y in zip(a, b))
    pick = lambda r: (r.value, r.ci_lower, r.ci_upper, int(r.n), int(r.converged), int(r.rounds), int(r.topup))
    for name in ("db", "db_mailbox"):
        o = g[name]
        assert o["refused"] is True
        want = (one.sum_amount(), one.avg_amount(), one.count_records(), one.sum_amount_where(250.0, 750.0))
        assert same_t(o["exact"], want, 1e-12), (name, o["exact"], want)
        for got, (agg, kw) in zip(o["approx"], FACADE_CALLS):
            assert same_t(got, pick(one.approx(agg, **kw)), 1e-9), (name, agg, kw, got)
        for got, w in zip(o["batch"], one.approx_batch(FACADE_BATCH)):
            assert same_t(got, pick(w), 1e-9), (name, got, pick(w))
        wg = one.approx_group_by("AVG", "product_id", 10.0)
        assert list(o["groups"]) == list(wg)
        for k, t in o["groups"].items():
            assert same_t(t, (wg[k].value, wg[k].ci_lower, wg[k].ci_upper, wg[k].n), 1e-8), (name, k)
        assert rel(o["psum"], one.parallel_sum_sample(5.0, seed=3)) <= 1e-12
        sch = aqe_backend.CustomApproximateScheduler(db=one, seed=77)
        ws = [(r.value, r.status.name, r.samples_used) for r in (sch.execute_sum_query("SELECT SUM(amount) FROM sales WHERE amount BETWEEN 100 AND 800", 5.0),
                                                                    sch.execute_avg_query("SELECT AVG(amount) FROM sales", 2.0), sch.execute_exact_sum())]
        assert all(a[1:] == b[1:] and rel(a[0], b[0]) <= 1e-12 for a, b in zip(o["sched"], ws)), (o["sched"], ws)
        assert abs(o["psum_unseeded"] - want[0]) / want[0] < 0.05
    numbers = lambda text: [float(x.replace(",", "")) for x in re.findall(r"(?<![\w.])-?\d[\d,]*\.\d+", "\n".join(l for l in text.splitlines() if "time" not in l))]
    for r, pr in enumerate(per_rank):
        for (rc, text), argv in zip(pr["cli"], FACADE_CLI):
            assert rc == 0
            if r:
                assert text == ""  # rank 0 reports
                continue
            assert f"{world} GPUs, sharded by row region" in text
            buf = io.StringIO()
            assert cli.run(cli.build_parser().parse_args(argv + ["--db", str(tmp_path / "va.db")]), buf) == 0
            a, b = numbers(text), numbers(buf.getvalue())
            assert len(a) == len(b) and len(a) >= 1 and all(rel(x, y) <= 1e-9 for x, y in zip(a, b)), (argv, text, buf.getvalue())
    one.close_database()


@pytest.mark.gpu
def test_cli_under_torchrun_shards_the_table(oracle, table, tmp_path):
    """The documented multi-GPU invocation — `python -m torch.distributed.run --nproc-per-node N -m approximatequeryengine_amd.cli …`
    — with two ranks sharing this box's GPU over gloo: every rank stages its region of the file, rank 0 prints one report, and
    its numbers are the single-GPU CLI's."""
    import io
    import re
    import subprocess
    import sys
    from approximatequeryengine_amd import cli
    from approximatequeryengine_amd.build import ROOT
    rows = table(200_003)
    p = tmp_path / "sales.db"
    assert oracle.file_write(p, rows) == 0
    env = dict(os.environ, PYTHONPATH=str(ROOT) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    numbers = lambda text: [float(x.replace(",", "")) for x in re.findall(r"(?<![\w.])-?\d[\d,]*\.\d+", "\n".join(l for l in text.splitlines() if "time" not in l))]
    # (--error / --sample spelled out: torchrun's own parser reads every option-like word first and takes `--e`, `--s` for
    # abbreviations of its own options)
    for argv in (["SELECT AVG(amount) FROM sales", "--error", "2"], ["SELECT SUM(amount) FROM sales WHERE amount BETWEEN 250 AND 750", "--sample", "5", "--method", "block", "--ci"]):
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                              "--master-port", str(_free_port()), "-m", "approximatequeryengine_amd.cli", *argv, "--db", str(p), "--backend", "gloo"],
                             env=env, capture_output=True, text=True, timeout=600, cwd=str(tmp_path))
        assert out.returncode == 0, out.stdout + out.stderr
        assert out.stdout.count("query: ") == 1 and "2 GPUs, sharded by row region" in out.stdout, out.stdout
        buf = io.StringIO()
        assert cli.run(cli.build_parser().parse_args(argv + ["--db", str(p)]), buf) == 0
        a, b = numbers(out.stdout), numbers(buf.getvalue())
        assert len(a) == len(b) and len(a) >= 2 and all(rel(x, y) <= 1e-9 for x, y in zip(a, b)), (out.stdout, buf.getvalue())


def _nccl_worker(port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from approximatequeryengine_amd.sharded_backend import ShardedBPlusDB
    pick = lambda r: (r.value, r.ci_lower, r.ci_upper, int(r.n), int(r.converged), int(r.rounds), int(r.topup))
    db = ShardedBPlusDB(device_id=0)
    assert db.open_database(os.path.join(out_dir, "va.db"))
    out = {"approx": [pick(db.approx(agg, **kw)) for agg, kw in FACADE_CALLS], "batch": [pick(r) for r in db.approx_batch(FACADE_BATCH)],
           "groups": {k: (g.value, g.ci_lower, g.ci_upper, g.n) for k, g in db.approx_group_by("AVG", "product_id", 10.0).items()}}
    db.close_database()
    torch.save(out, os.path.join(out_dir, "nccl.pt"))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
def test_sharded_facade_over_rccl_world_of_one(oracle, tmp_path):
    """The same façade with backend "nccl" (RCCL) — what a multi-GPU node runs — as far as one GPU can take it: a world of one
    rank, every collective of the sharded planners, queries, batches and GROUP BY going through RCCL on device tensors."""
    from approximatequeryengine_amd import aqe_backend
    assert oracle.file_write(tmp_path / "va.db", va_table(oracle, N_VA, 0, N_VA, ties=True)) == 0
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_worker, args=(_free_port(), str(tmp_path)))
    p.start()
    p.join(timeout=600)
    assert p.exitcode == 0
    got = torch.load(tmp_path / "nccl.pt", weights_only=False)
    one = aqe_backend.CustomBPlusDB()
    assert one.open_database(str(tmp_path / "va.db"))
    one._path = ""
    same_t = lambda a, b, tol: all((x == y) if isinstance(x, int) else rel(x, y) <= tol for x, y in zip(a, b))
    pick = lambda r: (r.value, r.ci_lower, r.ci_upper, int(r.n), int(r.converged), int(r.rounds), int(r.topup))
    for g, (agg, kw) in zip(got["approx"], FACADE_CALLS):
        assert same_t(g, pick(one.approx(agg, **kw)), 1e-9), (agg, kw, g)
    for g, w in zip(got["batch"], one.approx_batch(FACADE_BATCH)):
        assert same_t(g, pick(w), 1e-9)
    wg = one.approx_group_by("AVG", "product_id", 10.0)
    assert list(got["groups"]) == list(wg) and all(same_t(t, (wg[k].value, wg[k].ci_lower, wg[k].ci_upper, wg[k].n), 1e-8) for k, t in got["groups"].items())
    one.close_database()
