"""The Python mirror of the reference's `aqe_backend` module (bindings.cpp:10-137) and the CLI.
Host-only behaviour runs everywhere; anything that computes is marked gpu."""
import base64
import datetime
import io
import math

import numpy as np
import pytest

from helpers import digest, rel


def test_record_and_result_types_have_the_reference_fields():
    from approximatequeryengine_amd import aqe_backend as m
    r = m.Record()
    assert (r.id, r.amount, r.region, r.product_id, r.timestamp) == (0, 0.0, 0, 0, 0)  # DB.hpp:24
    r.id, r.amount = 7, 3.5
    assert m.Record(7, 3.5, 0, 0, 0) == r
    assert [s.name for s in m.CustomApproximationStatus] == ["STABLE", "DRIFTING", "INSUFFICIENT_DATA", "ERROR"]
    v = m.CustomValidationResult()
    for f in ("value", "status", "confidence_level", "error_margin", "samples_used", "computation_time"):  # SCH.hpp:15-22
        assert hasattr(v, f)
    assert isinstance(v.computation_time, datetime.timedelta)
    db = m.CustomBPlusDB()
    for name in ("create_database", "open_database", "close_database", "insert_record", "sum_amount", "sum_amount_where",
                 "sample_records", "get_total_records", "get_node_count", "save_to_file", "load_from_file",
                 "fast_pointer_sample", "slow_pointer_sample", "dual_pointer_sample", "parallel_pointer_sample",
                 "random_pointer_sample", "clt_validated_dual_pointer_sample", "optimized_clt_sample", "block_sample",
                 "page_sample", "parallel_block_sample", "memory_stride_sample", "optimized_address_arithmetic_sample",
                 "multithreaded_memory_stride_sample", "fast_aggregated_memory_stride_sum",
                 "random_start_memory_stride_sample", "direct_access_sample", "optimized_sequential_sample"):  # bindings.cpp:44-101
        assert callable(getattr(db, name)), name
    for name in ("index_based_sample", "node_skip_sample", "signal_based_clt_sample", "balanced_tree_sample"):
        with pytest.raises(NotImplementedError):
            getattr(db, name)(10.0)
    s = m.CustomApproximateScheduler()
    for name in ("create_database", "open_database", "close_database", "insert_record", "insert_batch", "execute_sum_query",
                 "execute_avg_query", "execute_count_query", "execute_exact_sum", "execute_exact_avg", "execute_exact_count",
                 "benchmark_query", "get_total_records", "get_tree_height", "get_database_size_mb"):  # bindings.cpp:103-123
        assert callable(getattr(s, name)), name


def test_host_side_bookkeeping_needs_no_gpu(tmp_path, oracle, golden):
    from approximatequeryengine_amd import aqe_backend as m
    db = m.CustomBPlusDB()
    assert db.create_database(str(tmp_path / "x.db")) is True
    for i in (5, 1, 3, 2, 4, 3):  # out of order + a duplicate id
        assert db.insert_record(m.Record(i, float(i), i % 4, i % 100, i)) is True
    assert db.get_total_records() == 6 and db.get_node_count() == 1 and db.get_tree_height() == 1
    assert list(db._leaf_order()["id"]) == [1, 2, 3, 3, 4, 5]
    # the reference's file, written by its own save_to_file, loads (its own load_from_file dead-locks)
    raw = base64.b64decode(golden["file_5_rows_b64"])
    p = tmp_path / "five.db"
    p.write_bytes(raw)
    db2 = m.CustomBPlusDB()
    assert db2.open_database(str(p)) is True and db2.get_total_records() == 5
    q = tmp_path / "again.db"
    assert db2.save_to_file(str(q)) is True and q.read_bytes() == raw
    assert db2.open_database(str(tmp_path / "missing.db")) is False
    assert m.CustomBPlusDB().memory_stride_sample(10.0) == []  # empty table -> empty list (DB.cpp:743-746)
    assert m.CustomBPlusDB().sum_amount() == 0.0


def test_cli_flag_semantics():
    from approximatequeryengine_amd import cli
    P = cli.build_parser()
    a = P.parse_args(["SELECT SUM(amount) FROM sales", "--s", "10"])
    assert a.s == 10 and cli.determine_query_type(a.query, a) == cli.QUERY_RANDOM      # the reference runs this EXACT
    a = P.parse_args(["SELECT SUM(amount) FROM sales", "--e", "2"])
    assert a.e == 2 and cli.determine_query_type(a.query, a) == cli.QUERY_CLT           # the reference rejects this
    a = P.parse_args(["SELECT SUM(amount) FROM sales", "--sample", "5", "--error", "1"])
    assert (a.s, a.e) == (5, 1)
    a = P.parse_args(["SELECT APPROX(AVG(amount)) FROM sales"])
    assert cli.determine_query_type(a.query, a) == cli.QUERY_EMBEDDED
    assert cli.parse_embedded_approx("SELECT APPROX(SUM(amount)) FROM sales") == ("SELECT SUM(amount) FROM sales", True)
    assert cli.parse_embedded_approx("select approx( count(*) ) from t")[0] == "select count(*) from t"
    a = P.parse_args(["SELECT COUNT(*) FROM sales"])
    assert cli.determine_query_type(a.query, a) == cli.QUERY_EXACT and cli.aggregate_of(a.query) == "COUNT"
    assert cli.get_optimal_method_for_query("SELECT SUM(amount) FROM t", 10_000_000) == "revolutionary"
    assert cli.get_optimal_method_for_query("SELECT SUM(amount) FROM t", 1000) == "clt"
    assert cli.get_optimal_method_for_query("SELECT AVG(amount) FROM t") == "random"
    out = io.StringIO()
    assert cli.run(P.parse_args(["--explain"]), out) == 0 and "clt" in out.getvalue()
    out = io.StringIO()
    assert cli.run(P.parse_args(["SELECT SUM(amount) FROM t", "--db", "/nonexistent.db"]), out) == 1


def test_sharded_facade_needs_a_process_group_and_the_cli_knows_its_flags():
    """sharded_backend.ShardedBPlusDB is CustomBPlusDB over a torch.distributed group (one process per GPU): outside one it says so;
    the CLI's --backend / --collective select how the ranks talk when it is launched under torchrun."""
    from approximatequeryengine_amd import cli
    from approximatequeryengine_amd.sharded_backend import ShardedBPlusDB
    from approximatequeryengine_amd import aqe_backend
    assert issubclass(ShardedBPlusDB, aqe_backend.CustomBPlusDB)
    with pytest.raises(RuntimeError, match="process group"):
        ShardedBPlusDB()
    a = cli.build_parser().parse_args(["SELECT AVG(amount) FROM sales", "--error", "2", "--backend", "gloo", "--collective", "mailbox"])
    assert (a.e, a.backend, a.collective) == (2, "gloo", "mailbox")
    a = cli.build_parser().parse_args(["SELECT AVG(amount) FROM sales"])
    assert (a.backend, a.collective) == ("nccl", "torch")


# ------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def db100k(table):
    from approximatequeryengine_amd import aqe_backend as m
    db = m.CustomBPlusDB()
    db.insert_array(table(100_000))
    yield db
    db.close_database()


@pytest.mark.gpu
def test_samplers_return_the_reference_rows_as_records(db100k, golden, table):
    rows = table(100_000)
    T = golden["tables"]["100000"]
    by = {(c["method"], c["pct"], tuple(c["args"])): c for c in T["calls"]}
    got = db100k.memory_stride_sample(1.0, 0)
    assert isinstance(got, list) and type(got[0]).__name__ == "Record"
    assert digest(np.array([r.id - 1 for r in got], dtype=np.uint64)) == by[("memory_stride_sample", 1.0, (0,))]["idx"]
    r0 = got[3]
    assert (r0.amount, r0.region, r0.product_id, r0.timestamp) == tuple(rows[r0.id - 1][k] for k in ("amount", "region", "product_id", "timestamp"))
    checks = [
        ("random_pointer_sample", (1.0, 42), ("random_pointer_sample", 1.0, (42,))),
        ("block_sample", (5.0, 1000), ("block_sample", 5.0, (1000,))),
        ("page_sample", (5.0, 4096), ("page_sample", 5.0, (4096,))),
        ("parallel_block_sample", (5.0, 300, 3), ("parallel_block_sample", 5.0, (300, 3))),
        ("optimized_clt_sample", (20.0, 0.95, 20, 7, 2.0), ("optimized_clt_sample", 20.0, (0.95, 20, 7, 2.0))),
        ("fast_pointer_sample", (5.0, 2), ("fast_pointer_sample", 5.0, (2,))),
        ("slow_pointer_sample", (5.0,), ("slow_pointer_sample", 5.0, ())),
        ("dual_pointer_sample", (20.0,), ("dual_pointer_sample", 20.0, ())),
        ("parallel_pointer_sample", (1.0, 4), ("parallel_pointer_sample", 1.0, (4,))),
        ("optimized_address_arithmetic_sample", (20.0,), ("optimized_address_arithmetic_sample", 20.0, ())),
    ]
    for name, args, key in checks:
        arr = getattr(db100k, name)(*args, as_array=True)
        assert digest((arr["id"] - 1).astype(np.uint64)) == by[key]["idx"], name
    arr = db100k.clt_validated_dual_pointer_sample(20.0, 0.95, 10, 4, 0.0, as_array=True)
    assert digest(np.sort(arr["id"] - 1).astype(np.uint64)) == by[("clt_validated_dual_pointer_sample", 20.0, (0.95, 10, 4, 0.0))]["idx"]


@pytest.mark.gpu
def test_exact_and_cpp_reducers(db100k, golden, oracle, table):
    T = golden["tables"]["100000"]
    assert rel(db100k.sum_amount(), T["exact"]["sum_amount"]) <= 1e-12
    assert rel(db100k.avg_amount(), T["exact"]["sum_amount"] / 100_000) <= 1e-12
    assert db100k.count_records() == 100_000
    w = T["exact"]["where"][0]
    assert rel(db100k.sum_amount_where(*w["range"]), w["sum"]) <= 1e-12
    rows = table(100_000)
    # parallel_*_sample draw their rows on the device (AQE_M_RANDOM_DEVICE; the reference shuffles with random_device,
    # DB.cpp:345-363): with a seed the sample is the keyed bijection's prefix, restated in tests/helpers.py
    from helpers import perm_rows
    idx = np.sort(perm_rows(100_000, 1.0, 9))
    m = oracle.moments_idx(rows, idx)
    assert rel(db100k.parallel_sum_sample(1.0, 4, seed=9), m.sum * 100.0) <= 1e-12        # DB.cpp:303
    assert rel(db100k.parallel_avg_sample(1.0, 4, seed=9), m.sum * 100.0 / 100_000) <= 1e-12
    assert db100k.parallel_count_sample(1.0, 4, seed=9) == 100_000
    mw = oracle.moments_idx(rows, idx, where=(250.0, 750.0))
    assert rel(db100k.parallel_sum_where_sample(250.0, 750.0, 1.0, 4, seed=9), mw.sum * 100.0) <= 1e-12
    ridx = oracle.idx_region_stride(100_000, 1.0, 4, 42)
    assert rel(db100k.fast_aggregated_memory_stride_sum(1.0, 4), oracle.moments_idx(rows, ridx).sum) <= 1e-12
    # unseeded: statistically sane (random_device in the reference)
    exact = T["exact"]["sum_amount"]
    assert abs(db100k.parallel_sum_sample(5.0) - exact) / exact < 0.05


@pytest.mark.gpu
def test_fused_approx_entry_points(db100k, oracle, table):
    rows = table(100_000)
    r = db100k.approx_sum(method="stride", sample_percent=1.0)
    idx = oracle.idx_memory_stride(100_000, 1.0)
    m = oracle.moments_idx(rows, idx)
    assert r.n == m.n and rel(r.value, m.sum * (100_000 / m.n)) <= 1e-12 and r.ci_lower < r.value < r.ci_upper
    r = db100k.approx_avg(method="clt", error_percent=2.0, round0=10, growth=1)   # reference cadence
    rc, want, _ = oracle.clt_run(rows, 15.0, 0.95, 10, 4, 2.0)                      # e=2 -> pct 15 (CLI:243-250)
    assert (r.n, r.converged, r.rounds, r.topup) == (want.final.n, want.converged, want.rounds, want.topup)
    assert rel(r.value, want.final.sum / want.final.n) <= 1e-12
    assert db100k.approx_count(method="stride", sample_percent=10.0).value == 100_000          # CLI:196-197
    r = db100k.approx_sum(method="block", sample_percent=5.0, where=(250.0, 750.0), convention="cpp")
    mw = oracle.moments_idx(rows, oracle.idx_block(100_000, 5.0, 1000), where=(250.0, 750.0))
    assert rel(r.value, mw.sum * 20.0) <= 1e-12
    from approximatequeryengine_amd import aqe_backend as mod
    with pytest.raises(RuntimeError, match="No samples collected"):
        mod.CustomBPlusDB().approx_sum(method="stride", sample_percent=1.0)
    with pytest.raises(RuntimeError, match="No samples collected"):
        db100k.approx_sum(method="stride", sample_percent=0.00001)
    # WHERE id BETWEEN 20001 AND 70000: key bounds prune the sampled index space
    r = db100k.approx_sum(method="stride", sample_percent=1.0, id_between=(20_001, 70_000))
    sub = rows[20_000:70_000]
    m = oracle.moments_idx(sub, oracle.idx_memory_stride(len(sub), 1.0))
    assert r.n == m.n and rel(r.value, m.sum * (len(sub) / m.n)) <= 1e-12
    got = db100k.random_start_memory_stride_sample(1.0, 0, seed=5, as_array=True)
    assert np.array_equal(got["id"] - 1, oracle.idx_random_start_stride(100_000, 1.0, 0, seed=5).astype(np.int64))


@pytest.mark.gpu
def test_scheduler_facade(table, golden):
    from approximatequeryengine_amd import aqe_backend as m
    s = m.CustomApproximateScheduler(seed=1)
    assert s.create_database("") is True
    s.insert_array(table(100_000))
    assert s.insert_record(100_001, 5.0, 1, 1, 100_000) is True
    exact = s.execute_exact_sum()
    assert exact.status == m.CustomApproximationStatus.STABLE and exact.confidence_level == 1.0 and exact.samples_used == 100_001
    assert rel(exact.value, golden["tables"]["100000"]["exact"]["sum_amount"] + 5.0) <= 1e-12
    r = s.execute_sum_query("SELECT SUM(amount) FROM sales", 10.0, 4)
    assert r.status == m.CustomApproximationStatus.STABLE and r.confidence_level == 0.95   # SCH.cpp:296-305
    assert r.error_margin == 0.1 and r.samples_used == int(100_001 * 10.0 / 100.0)         # SCH.cpp:70-72
    assert abs(r.value - exact.value) / exact.value < 0.03
    assert isinstance(r.computation_time, datetime.timedelta)
    w = s.execute_sum_query("SELECT SUM(amount) FROM sales WHERE amount BETWEEN 250 AND 750", 10.0)
    assert 0.3 < w.value / exact.value < 0.7
    assert abs(s.execute_avg_query("SELECT AVG(amount) FROM sales", 10.0).value - exact.value / 100_001) < 10
    assert s.execute_count_query("SELECT COUNT(*) FROM sales", 10.0).value == 100_000.0     # size_t(n*100/pct), DB.cpp:314
    b = s.benchmark_query("SUM", 10.0, 4)
    assert b.exact_value == exact.value and b.error_percentage < 3 and b.threads_used == 4
    assert s.get_total_records() == 100_001 and abs(s.get_database_size_mb() - 100_001 * 32 / 2**20) < 1e-9
    s.close_database()


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path, oracle, table):
    from approximatequeryengine_amd import cli
    rows = table(100_000)
    p = tmp_path / "sales.db"
    assert oracle.file_write(p, rows) == 0
    exact = math.fsum(rows["amount"])
    for argv, frag in (
        (["SELECT SUM(amount) FROM sales", "--db", str(p), "--s", "1", "--ci", "--compare"], "stride sampling (1.0%)"),
        (["SELECT AVG(amount) FROM sales", "--db", str(p), "--e", "2"], "CLT"),
        (["SELECT APPROX(SUM(amount)) FROM sales", "--db", str(p)], "CLT"),
        (["SELECT COUNT(*) FROM sales", "--db", str(p)], "exact"),
        (["SELECT AVG(amount) FROM sales GROUP BY region", "--db", str(p), "--s", "10", "--ci"], "GROUP BY region (rowid sample 10%)"),
        (["SELECT SUM(amount) FROM sales WHERE amount BETWEEN 250 AND 750 GROUP BY product_id", "--db", str(p)], "GROUP BY product_id (exact)"),
    ):
        out = io.StringIO()
        assert cli.run(cli.build_parser().parse_args(argv), out) == 0, out.getvalue()
        assert frag in out.getvalue(), out.getvalue()
    # the reference CLI's routing by table size (enhanced_aqe_cli.py:178-186): direct access above 10 k rows, sequential below
    for n_small, frag in ((20_000, "direct_access sampling (5.0%)"), (5_000, "sequential sampling (5.0%)")):
        ps = tmp_path / f"small{n_small}.db"
        assert oracle.file_write(ps, rows[:n_small]) == 0
        out = io.StringIO()
        assert cli.run(cli.build_parser().parse_args(["SELECT AVG(amount) FROM sales", "--db", str(ps), "--s", "5", "--ci"]), out) == 0, out.getvalue()
        assert frag in out.getvalue(), out.getvalue()
        val = float(out.getvalue().split("value:")[1].split()[0].replace(",", ""))
        assert abs(val - float(rows["amount"][:n_small].mean())) < 30.0
    out = io.StringIO()
    cli.run(cli.build_parser().parse_args(["SELECT SUM(amount) FROM sales", "--db", str(p)]), out)
    val = float(out.getvalue().split("value:")[1].split()[0].replace(",", ""))
    assert abs(val - exact) < 1e-3
    assert p.stat().st_size == 24 + 32 * 100_000  # the CLI never rewrites the database
    # scalar queries honour WHERE amount ... (the reference CLI drops it); the CLT sampler has no WHERE form and says so
    out = io.StringIO()
    cli.run(cli.build_parser().parse_args(["SELECT SUM(amount) FROM sales WHERE amount BETWEEN 250 AND 750", "--db", str(p)]), out)
    val = float(out.getvalue().split("value:")[1].split()[0].replace(",", ""))
    a = rows["amount"]
    assert abs(val - math.fsum(a[(a >= 250.0) & (a <= 750.0)])) < 1e-3
    out = io.StringIO()
    assert cli.run(cli.build_parser().parse_args(["SELECT AVG(amount) FROM sales WHERE amount > 500", "--db", str(p), "--e", "2"]), out) == 0
    assert "WHERE clause is ignored" in out.getvalue()


@pytest.mark.gpu
def test_approx_batch_is_approx_in_one_launch(db100k):
    """approx_batch: the keyword dictionaries of approx(), served by ONE launch; same answers as one call each."""
    specs = [dict(agg="AVG", method="clt", error_percent=2.0), dict(agg="SUM", method="clt", error_percent=0.0, round0=64, growth=2),
             dict(agg="SUM", method="stride", sample_percent=1.0), dict(agg="COUNT", method="block", sample_percent=5.0, where=(250.0, 750.0), convention="cpp"),
             dict(agg="AVG", method="exact"), dict(agg="SUM", method="random", sample_percent=1.0, seed=7),
             dict(agg="AVG", method="page", sample_percent=5.0, id_between=(1000, 60_000))]
    got = db100k.approx_batch(specs)
    for kw, g in zip(specs, got):
        w = db100k.approx(**kw)
        assert (g.n, g.visited, g.converged, g.rounds, g.topup) == (w.n, w.visited, w.converged, w.rounds, w.topup), kw
        assert rel(g.value, w.value) <= 1e-12 and rel(g.ci_lower, w.ci_lower) <= 1e-11 and g.method == kw["method"]


@pytest.mark.gpu
def test_group_by_entry_point(db100k, oracle, table):
    """approx_group_by: the reference's GroupResultWithCI shape (key string -> value, ci_lower, ci_upper)."""
    rows = table(100_000)
    # (rowid % 10 == 0 on this table only ever meets regions 1 and 3 — region = row % 4 — exactly as the reference's
    #  own sampler would; step 33 visits all four)
    assert list(db100k.approx_group_by("AVG", group_by="region", sample_percent=10)) == ["1", "3"]
    got = db100k.approx_group_by("AVG", group_by="region", sample_percent=3)
    want = oracle.group(rows, 1, sample_percent=3)
    assert list(got) == [str(k) for k, *_ in want] == ["0", "1", "2", "3"]
    for (k, n, s_, q_), g in zip(want, got.values()):
        v, lo, hi = oracle.group_ci(1, n, s_, q_, 3, reference_sum=True)  # AVG: the reference's own numbers
        value, ci_lower, ci_upper = g                                       # unpacks like executor.h's QueryResult
        assert g.n == n and abs(value - v) <= 1e-9 * v and abs(ci_lower - lo) <= 1e-8 * lo and abs(ci_upper - hi) <= 1e-8 * hi
    total = db100k.approx_group_by("SUM", group_by="product_id", method="exact", sample_percent=100)
    assert len(total) == 100 and abs(sum(g.value for g in total.values()) - math.fsum(rows["amount"])) <= 1e-6
    blocks = db100k.approx_group_by("COUNT", group_by="region", method="block", sample_percent=5, where=(250.0, 750.0))
    idx = oracle.idx_block(len(rows), 5.0, 1000)
    for k, n, *_ in oracle.group(rows, 1, idx=idx, where=(250.0, 750.0)):
        assert blocks[str(k)].n == n and blocks[str(k)].value == n * 20.0


@pytest.mark.gpu
def test_default_clt_cadence_on_a_big_table_doubles(oracle, table):
    """clt_validated_dual_pointer_sample with the reference's defaults on a table where the reference's cadence would
    mean tens of thousands of decision points: the mirror keeps the first check at check_interval and doubles."""
    from approximatequeryengine_amd import aqe_backend as m
    n = 2_000_000
    rows = table(n)
    db = m.CustomBPlusDB()
    db.insert_array(rows)
    got = db.clt_validated_dual_pointer_sample(20.0, as_array=True)          # conf 0.95, interval 10, 4 threads, 2 %
    rc, want, idx = oracle.clt_run(rows, 20.0, 0.95, 10, 4, 2.0, R0=10, growth=2, want_idx=True)
    assert rc == 0 and want.converged and len(got) == want.final.n
    assert np.array_equal(np.sort(got["id"] - 1), np.sort(idx.astype(np.int64)))
    small = m.CustomBPlusDB()                                                 # a small table keeps the cadence itself
    small.insert_array(rows[:50_000])
    got = small.clt_validated_dual_pointer_sample(20.0, as_array=True)
    rc, want, idx = oracle.clt_run(rows[:50_000], 20.0, 0.95, 10, 4, 2.0, R0=10, growth=1, want_idx=True)
    assert rc == 0 and len(got) == want.final.n and np.array_equal(np.sort(got["id"] - 1), np.sort(idx.astype(np.int64)))
    db.close_database()
    small.close_database()
