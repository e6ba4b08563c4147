"""GPU parity: every HIP kernel, called through the C ABI (libaqe_hip.so), against the oracle
(oracle/aqe_oracle.c) on the same seeded table, and against the golden vectors recorded from the
reference's own C++.

Bars: sample counts / index sets (via the gather kernels) bit-exact; sums 1e-12 relative (f64, different
summation order); estimates and interval bounds 1e-9 — far inside the 1e-6 the north star allows.
"""
import math

import numpy as np
import pytest

from helpers import digest, oracle_indices, rel

pytestmark = pytest.mark.gpu

SUM_TOL = 1e-12
EST_TOL = 1e-9


@pytest.fixture(scope="module")
def nat():
    from approximatequeryengine_amd import _native
    _native.lib()
    return _native


@pytest.fixture(scope="module")
def engines(nat, table):
    """engine(N) -> Engine with the seeded synthetic table of N rows staged from host rows (AoS kept)."""
    from approximatequeryengine_amd.engine import Engine
    cache = {}

    def get(n):
        if n not in cache:
            if len(cache) >= 3:
                k = next(iter(cache))
                cache.pop(k).close()
            e = Engine(0)
            e.stage_records(table(n), keep_aos=True)
            cache[n] = e
        return cache[n]

    yield get
    for e in cache.values():
        e.close()


def _query_for(nat, call):
    from approximatequeryengine_amd.engine import make_query
    m, pct, a = call["method"], call["pct"], call["args"]
    table_ = {
        "memory_stride_sample": lambda: make_query(nat.M_MEMORY_STRIDE, pct, stride_bytes=int(a[0])),
        "optimized_address_arithmetic_sample": lambda: make_query(nat.M_ADDRESS_ARITHMETIC, pct),
        "random_pointer_sample": lambda: make_query(nat.M_RANDOM_POINTER, pct, seed=int(a[0])),
        "direct_access_sample": lambda: make_query(nat.M_DIRECT_ACCESS, pct),
        "optimized_sequential_sample": lambda: make_query(nat.M_OPTIMIZED_SEQUENTIAL, pct, seed=int(a[0])),
        "block_sample": lambda: make_query(nat.M_BLOCK, pct, block_size=int(a[0])),
        "page_sample": lambda: make_query(nat.M_PAGE, pct, block_size=int(a[0])),
        "parallel_block_sample": lambda: make_query(nat.M_PARALLEL_BLOCK, pct, block_size=int(a[0]), num_threads=int(a[1])),
        "optimized_clt_sample": lambda: make_query(nat.M_OPTIMIZED_CLT, pct, num_threads=int(a[2])),
        "fast_pointer_sample": lambda: make_query(nat.M_FAST_POINTER, pct, step_size=int(a[0])),
        "slow_pointer_sample": lambda: make_query(nat.M_SLOW_POINTER, pct),
        "dual_pointer_sample": lambda: make_query(nat.M_DUAL_POINTER, pct),
        "parallel_pointer_sample": lambda: make_query(nat.M_PARALLEL_POINTER, pct, num_threads=int(a[0])),
        "adaptive_block_sample": lambda: make_query(nat.M_ADAPTIVE_BLOCK, pct, block_size=int(a[0]), block_size_max=int(a[1])),
        "stratified_block_sample": lambda: make_query(nat.M_STRATIFIED_BLOCK, pct, block_size=int(a[0]), num_threads=int(a[1])),
        "clt_validated_dual_pointer_sample": lambda: make_query(
            nat.M_CLT_DUAL_POINTER, pct, confidence_level=a[0], check_interval=int(a[1]), num_threads=int(a[2]),
            max_error_percent=a[3]),
    }
    return table_[m]()


def _check_against_oracle(nat, o, eng, rows, q, idx, where=None):
    """HIP aggregate of query q == oracle moments over the oracle's index list."""
    N = len(rows)
    res = eng.reduce(q)
    m = o.moments_idx(rows, idx, where=where)
    assert res.visited == len(idx)
    assert res.n == m.n
    if m.n:
        assert rel(res.sum, m.sum) <= SUM_TOL
        assert rel(res.sumsq, m.sumsq) <= SUM_TOL
        assert rel(res.m2, m.m2) <= 1e-9
        if where is None:
            est = o.lib.aqo_estimate_cli(q.agg, N, m.n, m.sum)
            assert rel(res.value, est) <= EST_TOL
            if m.n > 1 and q.agg != nat.COUNT:
                moe, lo, hi = o.ci_cli(q.agg, N, m.n, m.m2, est)
                assert rel(res.ci_lower, lo) <= EST_TOL and rel(res.ci_upper, hi) <= EST_TOL
    return res


@pytest.mark.parametrize("n", [10_000, 100_000, 100_007, 1_000_000, 10_000_000])
def test_golden_calls_reduce_and_gather(nat, oracle, golden, table, engines, n):
    """Every deterministic reference call in the golden file: the HIP reduce reproduces the sums the
    reference's samples give, and the HIP gather returns exactly the reference's rows."""
    T = golden["tables"][str(n)]
    rows, eng = table(n), engines(n)
    for call in T["calls"]:
        q = _query_for(nat, call)
        if call["method"] in ("memory_stride_sample", "optimized_address_arithmetic_sample") and "cache_rows" in T:
            q.visible_rows = T["cache_rows"]  # the reference's stale-cache length (DB.cpp:188-191)
        res = eng.reduce(q)
        g = call["idx"]
        assert res.visited == g["n"], call
        if g["n"]:
            assert rel(res.sum, call["fsum"]) <= SUM_TOL, call
            assert rel(res.sumsq, call["fsumsq"]) <= SUM_TOL, call
        # the same call through the other kernels (a multi-round plan takes the lean launch by default: the monitor-wave
        # kernel, and one launch per round): same counts, sums to rounding
        for other in (nat.Q_NO_LEAN, nat.Q_NO_PERSIST, nat.Q_FORCE_LEAN):  # (FORCE_LEAN: single-round samplers through k_sweep_lean at any size)
            q.flags = other
            alt = eng.reduce(q)
            assert (alt.visited, alt.n, alt.rounds, alt.converged, alt.topup) == (res.visited, res.n, res.rounds, res.converged, res.topup), call
            assert rel(alt.sum, res.sum) <= 1e-13 and rel(alt.sumsq, res.sumsq) <= 1e-13 and rel(alt.ci_lower, res.ci_lower) <= 1e-12, call
        q.flags = 0
        if "cli" in call and g["n"] > 1:
            c = call["cli"]
            assert rel(res.value, c["SUM"]) <= EST_TOL
            assert rel(res.ci_lower, c["SUM_ci"][0]) <= EST_TOL and rel(res.ci_upper, c["SUM_ci"][1]) <= EST_TOL
            q.agg = nat.AVG
            r2 = eng.reduce(q)
            assert rel(r2.value, c["AVG"]) <= EST_TOL
            assert rel(r2.ci_lower, c["AVG_ci"][0]) <= EST_TOL and rel(r2.ci_upper, c["AVG_ci"][1]) <= EST_TOL
            q.agg = nat.COUNT
            assert eng.reduce(q).value == c["COUNT"]
            q.agg = nat.SUM
        if "where" in call:
            w = call["where"]
            q.has_where, q.where_min, q.where_max = 1, w["range"][0], w["range"][1]
            rw = eng.reduce(q)
            assert rw.n == w["n"] and rw.visited == g["n"]
            assert rel(rw.sum, w["fsum"]) <= SUM_TOL
            q.has_where = 0
        # record-returning form: identical rows, identical order (CLT: as a multiset)
        got = eng.gather(q)
        idx = (got["id"] - 1).astype(np.uint64)
        if call.get("sorted"):
            idx = np.sort(idx)
        assert digest(idx) == g, call
        assert got.tobytes() == rows[(got["id"] - 1)].tobytes()


def test_golden_calls_at_one_hundred_million_rows(nat, oracle, golden):
    """BASELINE.json configs 2 and 4 at their own size (100 M rows): the reference's C++ run on the seeded table
    (oracle/make_golden_100m.py) against the HIP reduce on the same table generated on the device — sample counts,
    sums, the WHERE-filtered block sample — and, through the gather kernels, the reference's exact index sets."""
    from approximatequeryengine_amd.engine import Engine, make_query
    T = golden["tables"]["100000000"]
    n = T["N"]
    with Engine(0) as eng:
        eng.generate_synthetic(n, seed=42, keep_aos=True)  # (the generator is the oracle's, row for row: test_synthetic_generator_matches_oracle)
        ex = eng.reduce(make_query(nat.M_EXACT, 100.0))
        assert ex.n == n and rel(ex.value, T["exact_sum"]) <= SUM_TOL
        for call in T["calls"]:
            q = _query_for(nat, call)
            res = eng.reduce(q)
            g = call["idx"]
            assert res.visited == g["n"], call["method"]
            assert rel(res.sum, call["fsum"]) <= SUM_TOL and rel(res.sumsq, call["fsumsq"]) <= SUM_TOL, call["method"]
            if "where" in call:
                w = call["where"]
                q.has_where, q.where_min, q.where_max = 1, w["range"][0], w["range"][1]
                rw = eng.reduce(q)
                assert rw.n == w["n"] and rw.visited == g["n"] and rel(rw.sum, w["fsum"]) <= SUM_TOL
                q.has_where = 0
            got = eng.gather(q)
            assert digest((got["id"] - 1).astype(np.uint64)) == g, call["method"]


@pytest.mark.parametrize("n", [10_000, 1_000_000])
def test_seeded_random_pointer_seeds(nat, golden, table, engines, n):
    from approximatequeryengine_amd.engine import make_query
    eng = engines(n)
    for call in golden["tables"][str(n)]["random_seeds"]:
        q = make_query(nat.M_RANDOM_POINTER, call["pct"], seed=int(call["args"][0]))
        res = eng.reduce(q)
        assert res.visited == call["idx"]["n"] and rel(res.sum, call["fsum"]) <= SUM_TOL
        got = eng.gather(q)
        assert digest((got["id"] - 1).astype(np.uint64)) == call["idx"]


@pytest.mark.parametrize("n", [10_000, 100_007, 1_000_000])
def test_exact_scans(nat, oracle, golden, table, engines, n):
    from approximatequeryengine_amd.engine import make_query
    T, eng = golden["tables"][str(n)], engines(n)
    r = eng.reduce(make_query(nat.M_EXACT, 100.0, agg=nat.SUM))
    assert r.n == n and rel(r.value, T["exact"]["sum_amount"]) <= SUM_TOL
    assert r.ci_lower == r.value == r.ci_upper
    assert rel(eng.reduce(make_query(nat.M_EXACT, 100.0, agg=nat.AVG)).value, T["exact"]["sum_amount"] / n) <= SUM_TOL
    assert eng.reduce(make_query(nat.M_EXACT, 100.0, agg=nat.COUNT)).value == n
    for w in T["exact"]["where"]:
        rw = eng.reduce(make_query(nat.M_EXACT, 100.0, where=tuple(w["range"])))
        assert (rel(rw.value, w["sum"]) <= SUM_TOL) if w["sum"] else rw.value == 0.0
    # the same scans as lean launches (what scans of 12 MB and more take by themselves), and over a row window with odd ends
    rl = eng.reduce(make_query(nat.M_EXACT, 100.0, agg=nat.SUM, flags=nat.Q_FORCE_LEAN))
    assert (rl.n, rl.visited) == (n, n) and rel(rl.value, T["exact"]["sum_amount"]) <= SUM_TOL and rel(rl.sumsq, r.sumsq) <= SUM_TOL
    for w in T["exact"]["where"]:
        rw = eng.reduce(make_query(nat.M_EXACT, 100.0, where=tuple(w["range"]), flags=nat.Q_FORCE_LEAN))
        assert (rel(rw.value, w["sum"]) <= SUM_TOL) if w["sum"] else rw.value == 0.0
    lo, hi = 1 + n // 7, n - n // 5 - 1
    a = eng.reduce(make_query(nat.M_EXACT, 100.0, rows=(lo, hi)))
    b = eng.reduce(make_query(nat.M_EXACT, 100.0, rows=(lo, hi), flags=nat.Q_FORCE_LEAN))
    want = oracle.moments_idx(table(n), np.arange(lo, hi, dtype=np.uint64))
    assert a.n == b.n == want.n == hi - lo and rel(a.sum, want.sum) <= SUM_TOL and rel(b.sum, want.sum) <= SUM_TOL


CLT_CASES = [
    # (N, pct, conf, check_interval, T, e, R0, growth)
    (100_000, 20.0, 0.95, 10, 4, 2.0, 10, 1),      # reference cadence, converges (rule A) -> top-up
    (100_000, 20.0, 0.95, 10, 4, 0.0, 10, 1),      # never converges: full dual-pointer sweep
    (100_000, 20.0, 0.99, 10, 4, 1.0, 64, 2),      # geometric rounds
    (100_007, 10.0, 0.90, 20, 6, 3.0, 20, 1),
    (100_007, 5.0, 0.95, 4, 3, 0.5, 16, 3),
    (1_000_000, 20.0, 0.95, 10, 4, 1.0, 1024, 2),
    (1_000_000, 20.0, 0.95, 10, 4, 0.01, 4096, 2), # e = 0.01 % never converges at 1M
    (10_000, 33.3, 0.95, 10, 2, 5.0, 10, 1),
    (10_000, 100.0, 0.95, 10, 4, 0.0, 7, 1),
    (10_000, 20.0, 0.95, 10, 1, 2.0, 10, 1),       # T=1: no fast pointer at all
    (1_000_000, 20.0, 0.95, 10, 4, 1.0, 4096, 4),  # the bench schedule, early stop
    (1_000_000, 20.0, 0.95, 10, 4, 0.0, 4096, 4),  # the bench schedule, full sweep
    (1_000_000, 10.0, 0.95, 10, 8, 0.3, 16, 2),    # 15 rounds, stops in the middle, rounds abandoned
    (1_000_000, 20.0, 0.95, 10, 5, 0.2, 100, 3),   # odd thread count: fast and slow regions differ (no pairs)
    (100_007, 20.0, 0.95, 10, 64, 1.0, 8, 2),      # 64 pointers
    (1_000_000, 20.0, 0.95, 10, 4, 2.0, 10, 1),    # the reference's cadence and defaults: 5000 rounds planned, stops early
    (200_000, 20.0, 0.95, 10, 4, 0.0, 10, 1),      # ... and never converging: 1000 rounds, launched chunk by chunk
    (1_000_000, 20.0, 0.95, 10, 128, 0.0, 1024, 4),  # 128 pointers: 256 runs, the wide form of the lean launch
    (10_000_000, 20.0, 0.95, 10, 4, 0.01, 4096, 4),  # the bench query at its own size: 4 M samples, never converges
    (10_000_000, 20.0, 0.95, 10, 4, 1.0, 4096, 4),   # ... and its other reading: stops after round 0, 500 000-row top-up
]


@pytest.mark.parametrize("case", CLT_CASES, ids=lambda c: "N%d_p%g_T%d_e%g_R%d_g%d" % (c[0], c[1], c[4], c[5], c[6], c[7]))
def test_clt_monitor_matches_oracle(nat, oracle, table, engines, case):
    from approximatequeryengine_amd.engine import make_query
    n, pct, conf, ci, T, e, R0, g = case
    rows, eng = table(n), engines(n)
    rc, want, idx = oracle.clt_run(rows, pct, conf, ci, T, e, R0=R0, growth=g, want_idx=True)
    assert rc == 0
    q = make_query(nat.M_CLT_DUAL_POINTER, pct, agg=nat.AVG, confidence_level=conf, check_interval=ci,
                   num_threads=T, max_error_percent=e, clt_round0=R0, clt_growth=g)
    # the same query as one launch per round must agree with the single persistent launch, and with whichever
    # of the two the library picks by itself (it predicts whether the query stops early)
    q.flags = nat.Q_NO_PERSIST
    multi = eng.reduce(q)
    q.flags = 0
    auto = eng.reduce(q)
    assert (auto.n, auto.visited, auto.converged, auto.rounds, auto.topup) == (multi.n, multi.visited, multi.converged, multi.rounds, multi.topup)
    assert rel(auto.sum, multi.sum) <= 1e-14 and rel(auto.ci_lower, multi.ci_lower) <= 1e-13
    q.flags = nat.Q_FORCE_PERSIST | nat.Q_SHARE_GPU | nat.Q_NO_LAYOUT  # half the compute units, column swept in place
    alt = eng.reduce(q)
    assert (alt.n, alt.visited, alt.converged, alt.rounds, alt.topup) == (multi.n, multi.visited, multi.converged, multi.rounds, multi.topup)
    assert rel(alt.sum, multi.sum) <= 1e-14 and rel(alt.sumsq, multi.sumsq) <= 1e-14 and rel(alt.ci_lower, multi.ci_lower) <= 1e-13
    # ... and both single-launch kernels: the persistent sweep with its monitor wave, and the lean launch that judges every
    # round once at the end (taken by default where the whole sweep is in flight at once)
    q.flags = nat.Q_FORCE_PERSIST | nat.Q_NO_LEAN
    mon = eng.reduce(q)
    assert (mon.n, mon.visited, mon.converged, mon.rounds, mon.topup) == (multi.n, multi.visited, multi.converged, multi.rounds, multi.topup)
    assert rel(mon.sum, multi.sum) <= 1e-14 and rel(mon.sumsq, multi.sumsq) <= 1e-14 and rel(mon.ci_lower, multi.ci_lower) <= 1e-13
    q.flags = nat.Q_FORCE_PERSIST
    res = eng.reduce(q)
    # (different summation trees: integers and decisions identical, sums to rounding)
    assert (res.n, res.visited, res.converged, res.rounds, res.topup) == (multi.n, multi.visited, multi.converged, multi.rounds, multi.topup)
    assert rel(res.sum, multi.sum) <= 1e-14 and rel(res.sumsq, multi.sumsq) <= 1e-14 and rel(res.ci_lower, multi.ci_lower) <= 1e-13
    assert res.converged == want.converged, (res.converged, want.converged)
    assert res.rounds == want.rounds
    assert res.topup == want.topup
    assert res.n == want.final.n == res.visited
    assert rel(res.sum, want.final.sum) <= SUM_TOL
    assert rel(res.m2, want.final.m2) <= 1e-9
    est = want.final.sum / want.final.n
    assert rel(res.value, est) <= EST_TOL
    moe, lo, hi = oracle.ci_cli(nat.AVG, n, want.final.n, want.final.m2, est)
    assert rel(res.ci_lower, lo) <= EST_TOL and rel(res.ci_upper, hi) <= EST_TOL
    got = eng.gather(q)
    assert np.array_equal(np.sort(got["id"] - 1), np.sort(idx.astype(np.int64)))


def test_small_table_samplers_against_the_reference_fixture(nat, oracle):
    """direct_access_sample / optimized_sequential_sample (what the reference CLI takes below 50 k rows): the HIP reduce and
    gather against the reference's recorded rows (tests/golden/small_tables.json) and the oracle."""
    import json, os
    from approximatequeryengine_amd.engine import Engine, make_query
    G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "small_tables.json")))
    for n_s, T in G["tables"].items():
        n = int(n_s)
        rows = oracle.synth(n, G["seed"])
        with Engine(0) as eng:
            eng.stage_records(rows, keep_aos=True)
            for d in T["direct_access"]:
                q = make_query(nat.M_DIRECT_ACCESS, d["pct"])
                got = eng.gather(q)
                assert digest(got["id"] - 1) == d["idx"], (n, d["pct"])  # the reference's rows, in its order, duplicates included
                idx = oracle.idx_direct_access(n, d["pct"])
                if len(idx):
                    _check_against_oracle(nat, oracle, eng, rows, q, idx)
                    q.has_where, q.where_min, q.where_max = 1, 200.0, 640.0
                    _check_against_oracle(nat, oracle, eng, rows, q, idx, where=(200.0, 640.0))
            for pct, seed in ((1.0, 5), (12.5, 42), (37.5, 77)):
                q = make_query(nat.M_OPTIMIZED_SEQUENTIAL, pct, seed=seed)
                idx = oracle.idx_optimized_sequential(n, pct, seed)
                got = eng.gather(q)
                assert np.array_equal(got["id"] - 1, idx.astype(np.int64))
                if len(idx):
                    _check_against_oracle(nat, oracle, eng, rows, q, idx)


def test_clt_on_the_heterogeneous_table_of_the_reference_fixture(nat, oracle):
    """tests/golden/clt_hetero.json (reference runs on a table whose halves have spreads 1 : 0.2): the HIP path gives the
    restatement's answer recorded there — the leader's stop point, which IS the reference's where thread 0 converges first
    and is later than the reference's where another fast thread does (test_oracle_golden.py says how far)."""
    import json, os
    from approximatequeryengine_amd.engine import Engine, make_query
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "clt_hetero.json")))
    n = g["rows"]
    for flip in (False, True):
        rows = oracle.synth(n, g["seed"])
        a, b = g["scales"]
        rows["amount"] = 500.5 + (rows["amount"] - 500.5) * np.where(np.arange(n) < n // 2, b if flip else a, a if flip else b)
        with Engine(0) as eng:
            eng.stage_records(rows, keep_aos=False)
            for case in (c for c in g["cases"] if c["flip"] == flip):
                want = case["restatement"]
                for R0, growth in ((case["check_interval"], 1), (64, 2)):  # the reference's cadence; a geometric schedule against the oracle
                    q = make_query(nat.M_CLT_DUAL_POINTER, case["pct"], agg=nat.AVG, check_interval=case["check_interval"], num_threads=case["T"],
                                   max_error_percent=case["e"], clt_round0=R0, clt_growth=growth)
                    r = eng.reduce(q)
                    rc, w, _ = oracle.clt_run(rows, case["pct"], 0.95, case["check_interval"], case["T"], case["e"], R0=R0, growth=growth)
                    assert rc == 0
                    assert (r.n, r.topup, r.converged, r.rounds) == (w.final.n, w.topup, w.converged, w.rounds), (case["T"], case["e"], R0)
                    assert rel(r.sum, w.final.sum) <= SUM_TOL and rel(r.value, w.final.sum / w.final.n) <= EST_TOL
                    if growth == 1:
                        assert (r.n, r.topup, r.converged, r.rounds) == (want["n"], want["topup"], want["converged"], want["rounds"])
                        assert rel(r.value, want["avg"]) <= EST_TOL


@pytest.mark.parametrize("n", [1_000_000, 10_000_000])
def test_clt_stop_points_of_the_reference(nat, golden, engines, n):
    """Where the CLT monitor stops, against the reference's own runs (golden clt_fast_stop, distributions), at 1 M rows and
    at the bench's 10 M.  T = 2 — the race-free regime: ONE fast thread, whose own statistics decide (DB.cpp:936-961) —
    the device's leader must stop on exactly the row count the reference's fast thread had taken.  T = 4 at e = 1 % —
    the racy regime — the rows collected before the stop must lie within what the reference's 30 recorded runs show
    (they equal T x the fast thread's count: every worker takes the same number of rows per round)."""
    from approximatequeryengine_amd.engine import make_query
    T = golden["tables"][str(n)]
    eng = engines(n)
    for g in T["clt_fast_stop"]:
        q = make_query(nat.M_CLT_DUAL_POINTER, g["pct"], agg=nat.AVG, confidence_level=0.95, check_interval=g["check_interval"], num_threads=2,
                       max_error_percent=g["e"])
        r = eng.reduce(q)
        assert r.converged == 1 and r.rounds == g["n_fast_at_stop"] // g["check_interval"], (g, r.rounds)
        assert r.n - r.topup == 2 * g["n_fast_at_stop"]  # the leader and the one slow pointer, row for row
    runs = T["distributions"]["clt_e1_pct20_T4"]
    base4 = int(n * 0.2) // 4
    r = eng.reduce(make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, confidence_level=0.95, check_interval=10, num_threads=4, max_error_percent=1.0))
    ref_collected = [x["n"] - base4 for x in runs]
    assert r.converged == 1 and r.topup == base4
    assert min(ref_collected) <= r.n - r.topup <= max(ref_collected), (r.n - r.topup, min(ref_collected), max(ref_collected))
    avgs = np.array([x["avg"] for x in runs])
    assert abs(r.value - avgs.mean()) <= 3 * max(avgs.std(), 0.5)


def test_clt_head_form_misprediction_is_continued(nat, oracle, table):
    """A query predicted to stop early runs as ONE small launch over the first rounds (the head form).  When the
    prediction fails — here the table's head is nearly constant, the rest is not — fetch() launches the remaining
    rounds, and the plan takes the full single launch from then on: all three agree with the round-by-round form
    and with the oracle (DB.cpp:885-1043 semantics)."""
    from approximatequeryengine_amd.engine import Engine, make_query
    n = 1_000_000
    rows = table(n).copy()
    rows["amount"][:4096] = 500.0 + 1e-3 * (np.arange(4096) % 7)
    eng = Engine(0)
    try:
        eng.stage_records(rows, keep_aos=True)
        for e, R0, g in ((1.0, 256, 2), (0.05, 1024, 4), (1.0, 4096, 4)):
            rc, want, _ = oracle.clt_run(rows, 20.0, 0.95, 10, 4, e, R0=R0, growth=g, want_idx=True)
            assert rc == 0
            q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, num_threads=4, max_error_percent=e, clt_round0=R0, clt_growth=g)
            q.flags = nat.Q_NO_PERSIST
            multi = eng.reduce(q)
            q.flags = 0
            for _ in range(3):  # head + continuation, then the full launch twice
                r = eng.reduce(q)
                assert (r.n, r.visited, r.converged, r.rounds, r.topup, r.topup_pending) == (multi.n, multi.visited, multi.converged, multi.rounds, multi.topup, 0)
                assert rel(r.sum, multi.sum) <= 1e-14 and rel(r.ci_lower, multi.ci_lower) <= 1e-13 and rel(r.value, multi.value) <= 1e-13
            assert (r.converged, r.rounds, r.topup, r.n) == (want.converged, want.rounds, want.topup, want.final.n)
            assert rel(r.sum, want.final.sum) <= SUM_TOL
    finally:
        eng.close()


def test_clt_decision_table_near_the_thresholds(nat, oracle, engines):
    """The device decides both rules (DB.cpp:936-961, 993-1016) by comparisons cleared of divisions and falls back
    to the reference's expressions inside a guard band.  Crafted moment vectors are folded through the stepwise
    API (k_update) at relative distances 1e-2 ... 1e-15 on either side of each threshold: well outside the band the
    oracle's decision function is the expectation; everywhere the literal expressions, evaluated on the very
    moments the device sees, are."""
    import torch
    from approximatequeryengine_amd.engine import make_query
    eng = engines(100_000)
    c = eng.info().shift
    z = oracle.lib.aqo_clt_zscore(0.95)

    def literal(n_a, sd_a, qd_a, n_b, sd_b, qd_b, e, base):
        n, sd, qd = n_a, sd_a, qd_a  # rule A: the leader's own samples (group a), DB.cpp:936-961
        if n >= 30:
            mean, m2 = c + sd / n, max(qd - sd * sd / n, 0.0)
            err = (z * math.sqrt(m2 / (n - 1.0) / n) / mean) * 100.0
            if err <= e and n >= 50:
                return 1
        if n_b >= 20 and n_a >= 30:
            ma, mb = c + sd_a / n_a, c + sd_b / n_b
            if ma > 0 and abs(mb - ma) / ma <= e / 100.0 and n_a >= base // 2:
                return 2
        return 0

    def vec(n_a, mean_a, var_a, n_b, mean_b, var_b):
        out = []
        for n, mean, var in ((n_a, mean_a, var_a), (n_b, mean_b, var_b)):
            sd = n * (mean - c)
            out += [float(n), sd, var * (n - 1) + sd * sd / n if n else 0.0]
        return out + [float(n_a + n_b), 0.0]

    e = 2.0
    q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=e, clt_round0=1024, clt_growth=2)
    plan = eng.plan(q)
    base = int(100_000 * 20.0 / 100)
    dev = torch.zeros(nat.MOMENT_VEC, dtype=torch.float64, device="cuda")
    cases = []
    for n in (40, 60, 5000):
        mean = 500.0
        var0 = (e * mean / (100.0 * z)) ** 2 * n  # err == e exactly (in exact arithmetic)
        for d in (0.0, 1e-15, 1e-13, 1e-10, 1e-8, 1e-5, 1e-2):
            for sgn in (-1.0, 1.0):
                cases.append(("A", d, vec(n, mean, var0 * (1 + sgn * d), n // 3, mean * 1.5, var0)))  # (group b must not matter)
    for n_a in (base // 2 - 1, base // 2, base):
        mean_a = 480.0
        for d in (0.0, 1e-15, 1e-12, 1e-9, 1e-5, 1e-2):
            for sgn in (-1.0, 1.0):
                mean_b = mean_a * (1 + (e / 100.0) * (1 + sgn * d))
                cases.append(("B", d, vec(n_a, mean_a, 1e7, 25, mean_b, 1e7)))  # huge variance: rule A stays off
    cases.append(("neg", 1.0, vec(80, -5.0, 1.0, 40, -5.0, 1.0)))  # mean < 0: err < 0 <= e in the reference's expression
    seen = set()
    for kind, d, v in cases:
        dev.copy_(torch.tensor(v, dtype=torch.float64))
        torch.cuda.synchronize()
        plan.enqueue_update(0, dev.data_ptr())
        plan.enqueue_finalize()
        got = plan.fetch().converged
        assert got == literal(*v[:6], e, base), (kind, d, v)
        seen.add((kind, got))
        if kind == "A" and d >= 1e-5:
            n = int(v[0])
            mean, m2 = c + v[1] / n, v[2] - v[1] ** 2 / n
            assert (got == 1) == bool(oracle.lib.aqo_clt_fast_rule(n, mean, m2 / (n - 1), z, e)) or n < 50
    assert {("A", 0), ("A", 1), ("B", 0), ("B", 2), ("neg", 1)} <= seen
    plan.close()


def test_fetch_polls_the_pinned_result_while_the_stream_is_busy(nat, engines):
    """aqe_plan_fetch takes a result from the plan's pinned block as soon as the finishing launch has written all of
    it (check word over every field, kernels.hpp result_check) — also while later launches of other plans are still
    queued on the same stream, and in any order."""
    import torch
    from approximatequeryengine_amd.engine import make_query
    eng = engines(1_000_000)
    qs = [make_query(nat.M_MEMORY_STRIDE, 1.0), make_query(nat.M_BLOCK, 5.0, where=(250.0, 750.0)),
          make_query(nat.M_RANDOM_POINTER, 1.0, seed=7),
          make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.0, clt_round0=1024, clt_growth=4),
          make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=1024, clt_growth=4)]
    want = [eng.reduce(q) for q in qs]
    plans = [eng.plan(q) for q in qs]
    st = torch.cuda.Stream().cuda_stream
    key = lambda r: (r.n, r.visited, r.converged, r.rounds, r.topup, r.topup_pending, r.device_status)
    for it in range(60):
        order = list(range(len(plans)))
        if it % 2:
            order.reverse()
        for i in order:
            plans[i].enqueue_all(st)
        for i in (reversed(order) if it % 3 == 0 else order):
            r = plans[i].fetch(st)
            assert key(r) == key(want[i]), (it, i)
            assert rel(r.sum, want[i].sum) <= 1e-14 and rel(r.value, want[i].value) <= 1e-13 and rel(r.ci_upper, want[i].ci_upper) <= 1e-13
    for p in plans:
        p.close()


def test_plans_inherit_scratch_from_destroyed_plans(nat, engines):
    """A destroyed plan hands its device scratch (partials, counters, state, pinned result block, control block) to the
    next plan of the context as it is — whatever kind of query either was.  Plans of five kinds are created, run
    once (fused form; the CLT ones also through the stepwise per-round API) and destroyed in rotation."""
    import torch
    from approximatequeryengine_amd.engine import make_query
    eng = engines(1_000_000)
    qs = [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=512, clt_growth=2),   # stops, top-up
          make_query(nat.M_MEMORY_STRIDE, 2.0),
          make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.0, clt_round0=4096, clt_growth=4),  # runs to the end
          make_query(nat.M_BLOCK, 3.0, where=(100.0, 400.0)),
          make_query(nat.M_RANDOM_POINTER, 0.5, seed=3)]
    want = [eng.reduce(q) for q in qs]
    key = lambda r: (r.n, r.visited, r.converged, r.rounds, r.topup, r.topup_pending, r.device_status)
    vec = torch.zeros(nat.MOMENT_VEC, dtype=torch.float64, device="cuda")
    for it in range(40):
        for i, q in enumerate(qs):
            p = eng.plan(q)
            p.enqueue_all()
            r = p.fetch()
            assert key(r) == key(want[i]), (it, i)
            assert rel(r.sum, want[i].sum) <= 1e-14 and rel(r.value, want[i].value) <= 1e-13
            if i in (0, 2) and it % 8 == 0:  # the same plan again, round by round (world of one: no reduction in between)
                for rnd in range(p.rounds + (1 if p.has_topup else 0)):
                    p.enqueue_round(rnd, vec.data_ptr())
                    p.enqueue_update(rnd, vec.data_ptr())
                p.enqueue_finalize()
                r = p.fetch()
                assert key(r) == key(want[i]), ("stepwise", it, i)
            p.close()


def test_randomised_sampler_parameters_against_the_oracle(nat, oracle):
    """Seeded sweep over table sizes and sampler parameters nobody picked by hand: for every case the HIP reduce
    equals the oracle's moments over the oracle's index list (with and without WHERE), and the HIP gather returns
    exactly those rows."""
    from approximatequeryengine_amd.engine import Engine
    import os
    rng = np.random.default_rng(int(os.environ.get("AQE_FUZZ_SEED", "20251004")))  # (the environment widens the sweep for a one-off hunt)
    pick = lambda *xs: xs[int(rng.integers(len(xs)))]
    cases = 0
    for _ in range(int(os.environ.get("AQE_FUZZ_TABLES", "14"))):
        n = int(pick(1, 2, 63, 64, 65, 999, 1000, 1001, 4097, 12_345, 99_999, 250_001, int(rng.integers(2000, int(os.environ.get("AQE_FUZZ_MAXN", "400000"))))))
        rows = oracle.synth(n, seed=int(rng.integers(1, 1000)))
        eng = Engine(0)
        try:
            eng.stage_records(rows, keep_aos=True)
            for _ in range(12):
                pct = float(pick(0.01, 0.37, 1.0, 3.3, 10.0, 25.0, 50.0, 99.0, 100.0, round(float(rng.uniform(0.05, 60.0)), 3)))
                m = pick("memory_stride_sample", "optimized_address_arithmetic_sample", "block_sample", "page_sample", "parallel_block_sample",
                         "optimized_clt_sample", "fast_pointer_sample", "slow_pointer_sample", "dual_pointer_sample", "parallel_pointer_sample",
                         "random_pointer_sample", "clt_validated_dual_pointer_sample", "direct_access_sample", "optimized_sequential_sample")
                args = {"memory_stride_sample": [int(pick(0, 32, 64, 96, 320, 3200, 32 * int(rng.integers(1, 500))))],
                        "optimized_address_arithmetic_sample": [],
                        "block_sample": [int(pick(1, 7, 100, 1000, 4096, int(rng.integers(1, 6000))))],
                        "page_sample": [int(pick(32, 512, 4096, 8192, 32 * int(rng.integers(1, 400))))],
                        "parallel_block_sample": [int(pick(10, 1000, int(rng.integers(1, 3000)))), int(rng.integers(1, 9))],
                        "optimized_clt_sample": [0.95, 20, int(rng.integers(1, 9))],
                        "fast_pointer_sample": [int(rng.integers(1, 8))],
                        "slow_pointer_sample": [], "dual_pointer_sample": [],
                        "parallel_pointer_sample": [int(rng.integers(1, 9))],
                        "random_pointer_sample": [int(rng.integers(0, 2**31 - 1))],
                        "direct_access_sample": [], "optimized_sequential_sample": [int(rng.integers(0, 2**31 - 1))],
                        "clt_validated_dual_pointer_sample": [float(pick(0.9, 0.95, 0.99)), int(pick(4, 10, 25)), int(pick(1, 2, 4, 6)), float(pick(0.0, 0.5, 2.0, 5.0))]}[m]
                if m == "clt_validated_dual_pointer_sample" and n > 60_000:
                    continue  # (the reference's cadence: tens of thousands of rounds on a big table; covered by CLT_CASES)
                if m == "random_pointer_sample":
                    pct = min(pct, 10.0)
                call = {"method": m, "pct": pct, "args": args}
                try:
                    idx = oracle_indices(oracle, rows, call)
                except AssertionError:
                    continue  # parameters the reference itself rejects (e.g. a CLT target below one row per pointer)
                if idx is None:
                    continue
                q = _query_for(nat, call)
                try:
                    _check_against_oracle(nat, oracle, eng, rows, q, idx)
                except nat.AqeError as e:
                    assert e.status == nat.ERR_INVALID and len(idx) == 0, (n, call, str(e))
                    continue
                if m != "clt_validated_dual_pointer_sample":  # (the CLT sampler has no WHERE form in the reference)
                    lo = float(rng.uniform(1.0, 600.0))
                    q.has_where, q.where_min, q.where_max = 1, lo, lo + float(rng.uniform(0.0, 500.0))
                    _check_against_oracle(nat, oracle, eng, rows, q, idx, where=(q.where_min, q.where_max))
                    # the same sample through the lean launch (single-round samplers take it from 12 MB on by themselves:
                    # runs, rows of blocks whose first and last block the window cuts, odd lengths, tiles dealt out wave by wave)
                    q.flags |= nat.Q_FORCE_LEAN
                    _check_against_oracle(nat, oracle, eng, rows, q, idx, where=(q.where_min, q.where_max))
                    q.has_where = 0
                    _check_against_oracle(nat, oracle, eng, rows, q, idx)
                    q.flags &= ~nat.Q_FORCE_LEAN
                got = eng.gather(q)
                assert np.array_equal(np.sort(got["id"] - 1), np.sort(np.asarray(idx, dtype=np.int64))), (n, call)
                cases += 1
        finally:
            eng.close()
    assert cases >= 100


def test_randomised_clt_schedules_against_the_oracle(nat, oracle):
    """Seeded sweep over CLT queries nobody picked by hand — table size, distribution (uniform, skewed, constant head),
    sample percent, pointers, error bound, round schedule — in whatever form the library picks (full launch, head form
    with or without the top-up slot, continuation after a failed prediction, launch by launch, graph): decisions,
    counts and sums equal the oracle's round-synchronous restatement of DB.cpp:885-1043."""
    import os
    from approximatequeryengine_amd.engine import Engine, make_query
    rng = np.random.default_rng(int(os.environ.get("AQE_FUZZ_SEED", "7")))
    pick = lambda *xs: xs[int(rng.integers(len(xs)))]
    cases = 0
    for _ in range(int(os.environ.get("AQE_FUZZ_TABLES", "8"))):
        n = int(pick(5_000, 20_000, 100_003, 400_000, 1_000_000, int(rng.integers(3_000, 700_000))))
        rows = oracle.synth(n, seed=int(rng.integers(1, 1000)))
        kind = pick("uniform", "skewed", "flat_head")
        if kind == "skewed":
            rows["amount"] = np.exp(rng.normal(3.0, 1.2, n))
        elif kind == "flat_head":
            # (noise, not a pattern: a periodic head lets a strided pointer draw one single value, and with zero variance
            # "error <= 0" hangs on the last bit of the reference's two-pass mean)
            rows["amount"][: min(n, 4096)] = 500.0 + 1e-3 * rng.random(min(n, 4096))
        eng = Engine(0)
        try:
            eng.stage_records(rows, keep_aos=False)
            for _ in range(6):
                pct = float(pick(5.0, 10.0, 20.0, 33.0, 50.0))
                conf = float(pick(0.9, 0.95, 0.99))
                ci = int(pick(4, 10, 20))
                T = int(pick(1, 2, 3, 4, 5, 6, 8, 16))
                e = float(pick(0.0, 0.05, 0.3, 1.0, 2.0, 5.0, 20.0))
                R0 = int(pick(16, 64, 256, 1024, 4096, int(rng.integers(8, 5000))))
                g = int(pick(2, 3, 4))
                rc, want, _ = oracle.clt_run(rows, pct, conf, ci, T, e, R0=R0, growth=g)
                q = make_query(nat.M_CLT_DUAL_POINTER, pct, agg=nat.AVG, confidence_level=conf, check_interval=ci, num_threads=T,
                               max_error_percent=e, clt_round0=R0, clt_growth=g)
                if rc != 0:  # parameters the reference divides by zero on
                    with pytest.raises(nat.AqeError):
                        eng.reduce(q)
                    continue
                for rep in range(2):  # (the second execution may take another form: a failed prediction switches the plan)
                    r = eng.reduce(q)
                    assert (r.converged, r.rounds, r.topup, r.n, r.visited, r.topup_pending) == \
                        (want.converged, want.rounds, want.topup, want.final.n, want.final.n, 0), (n, kind, pct, conf, ci, T, e, R0, g, rep)
                    assert rel(r.sum, want.final.sum) <= SUM_TOL and rel(r.m2, want.final.m2) <= 1e-9
                cases += 1
        finally:
            eng.close()
    assert cases >= 30


def test_clt_invalid_parameters_are_errors_not_crashes(nat, engines):
    """Where the reference divides by zero (DB.cpp:927, 985, 993) the C ABI returns AQE_ERR_INVALID."""
    from approximatequeryengine_amd.engine import make_query
    eng = engines(10_000)
    for kw in (dict(sample_percent=0.01, num_threads=4), dict(sample_percent=10.0, check_interval=1)):
        q = make_query(nat.M_CLT_DUAL_POINTER, **kw)
        with pytest.raises(nat.AqeError) as ei:
            eng.reduce(q)
        assert ei.value.status == nat.ERR_INVALID
    with pytest.raises(nat.AqeError):
        eng.reduce(make_query(nat.M_DUAL_POINTER, 0.02))
    for bad in (dict(agg=7), dict(convention=-1), dict(where=(float("nan"), 1.0))):
        with pytest.raises(nat.AqeError) as ei:
            eng.reduce(make_query(nat.M_MEMORY_STRIDE, 1.0, **bad))
        assert ei.value.status == nat.ERR_INVALID


def test_conventions_and_where_composition(nat, oracle, table, engines):
    """C++ convention (DB.cpp:303-315), raw sum (DB.cpp:2046) and block ∘ WHERE ∘ scale (config 5)."""
    from approximatequeryengine_amd.engine import make_query
    n = 1_000_000
    rows, eng = table(n), engines(n)
    idx = oracle.idx_block(n, 1.0, 1000)
    for where in (None, (250.0, 750.0), (0.0, 10.0), (2000.0, 3000.0)):
        m = oracle.moments_idx(rows, idx, where=where)
        for agg in (nat.SUM, nat.AVG, nat.COUNT):
            q = make_query(nat.M_BLOCK, 1.0, agg=agg, convention=nat.EST_CPP, where=where, block_size=1000)
            r = eng.reduce(q)
            assert r.n == m.n and r.visited == len(idx)
            want = oracle.lib.aqo_estimate_cpp(agg, n, 1.0, len(idx), m.sum)
            assert (rel(r.value, want) <= EST_TOL) if want else r.value == 0.0
        r = eng.reduce(make_query(nat.M_BLOCK, 1.0, convention=nat.EST_RAW, where=where, block_size=1000))
        assert (rel(r.value, m.sum) <= SUM_TOL) if m.n else r.value == 0.0
    # region-stride reducer: fast_aggregated_memory_stride_sum semantics with seeded starts
    for seed in (1, 7, 42):
        idx = oracle.idx_region_stride(n, 1.0, 4, seed)
        r = eng.reduce(make_query(nat.M_REGION_STRIDE, 1.0, convention=nat.EST_RAW, num_threads=4, seed=seed))
        assert r.n == len(idx) and rel(r.value, oracle.moments_idx(rows, idx).sum) <= SUM_TOL


def test_key_range_windows(nat, oracle, table):
    """`WHERE id BETWEEN a AND b`: B+-tree key bounds -> row window -> the sampler runs on the window."""
    from approximatequeryengine_amd.engine import Engine, make_query
    n = 100_000
    rows = table(n)
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=False)             # dense ids (id = row + 1): arithmetic bounds
        assert eng.key_range_rows(1, n) == (0, n)
        assert eng.key_range_rows(5001, 25_000) == (5000, 25_000)
        assert eng.key_range_rows(-50, 10) == (0, 10) and eng.key_range_rows(n + 5, n + 9) == (0, 0)
        lo, hi = eng.key_range_rows(20_001, 70_000)
        sub = rows[lo:hi]
        for q, idx in (
            (make_query(nat.M_MEMORY_STRIDE, 1.0, rows=(lo, hi)), oracle.idx_memory_stride(len(sub), 1.0)),
            (make_query(nat.M_BLOCK, 10.0, rows=(lo, hi), where=(250.0, 750.0)), oracle.idx_block(len(sub), 10.0, 1000)),
            (make_query(nat.M_RANDOM_POINTER, 2.0, seed=3, rows=(lo, hi)), oracle.idx_random_pointer(len(sub), 2.0, 3)),
            (make_query(nat.M_RANDOM_START_STRIDE, 1.0, seed=11, rows=(lo, hi)), oracle.idx_random_start_stride(len(sub), 1.0, 0, seed=11)),
            (make_query(nat.M_EXACT, 100.0, rows=(lo, hi)), np.arange(len(sub), dtype=np.uint64)),
        ):
            r = eng.reduce(q)
            m = oracle.moments_idx(sub, idx, where=(250.0, 750.0) if q.has_where else None)
            assert (r.visited, r.n) == (len(idx), m.n) and rel(r.sum, m.sum) <= SUM_TOL
            if not q.has_where and q.method != nat.M_EXACT:  # SUM scales to the WINDOW's row count
                assert rel(r.value, m.sum * (len(sub) / m.n)) <= EST_TOL
        rc, want, _ = oracle.clt_run(sub, 20.0, 0.95, 10, 4, 1.0, R0=64, growth=2)
        r = eng.reduce(make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=64, clt_growth=2, rows=(lo, hi)))
        assert (r.n, r.converged, r.rounds, r.topup) == (want.final.n, want.converged, want.rounds, want.topup)
        assert rel(r.sum, want.final.sum) <= SUM_TOL
        # sparse ids: bounds come from a bisection over the resident rows
        sparse = rows.copy()
        sparse["id"] = 7 + 3 * np.arange(n)
        eng.stage_records(sparse, keep_aos=True)
        ids = sparse["id"]
        for a, b in ((7, 7), (8, 9), (100, 1000), (0, 10**9), (ids[-1], ids[-1] + 5), (ids[500] + 1, ids[900] - 1)):
            assert eng.key_range_rows(int(a), int(b)) == (int(np.searchsorted(ids, a, "left")), int(np.searchsorted(ids, b, "right")))
        eng.stage_records(sparse, keep_aos=False)
        with pytest.raises(nat.AqeError):
            eng.key_range_rows(10, 20)


def test_edge_shapes(nat, oracle, table):
    """Empty and tiny tables, 100 % samples, ragged tails, block sizes that do not divide N."""
    from approximatequeryengine_amd.engine import Engine, make_query
    for n in (0, 1, 63, 64, 65, 511, 513, 1000, 4097):
        rows = table(max(n, 1))[:n]
        with Engine(0) as eng:
            eng.stage_records(rows, keep_aos=True)
            for q, idx in (
                (make_query(nat.M_MEMORY_STRIDE, 50.0), oracle.idx_memory_stride(n, 50.0)),
                (make_query(nat.M_MEMORY_STRIDE, 100.0), oracle.idx_memory_stride(n, 100.0)),
                (make_query(nat.M_BLOCK, 30.0, block_size=7), oracle.idx_block(n, 30.0, 7)),
                (make_query(nat.M_PAGE, 100.0, block_size=4096), oracle.idx_page(n, 100.0, 4096)),
                (make_query(nat.M_RANDOM_POINTER, 100.0, seed=3), oracle.idx_random_pointer(n, 100.0, 3)),
                (make_query(nat.M_PARALLEL_BLOCK, 60.0, block_size=5, num_threads=3), oracle.idx_parallel_block(n, 60.0, 5, 3)),
                (make_query(nat.M_EXACT, 100.0), np.arange(n, dtype=np.uint64)),
            ):
                r = eng.reduce(q)
                assert r.visited == len(idx), (n, q.method)
                if len(idx):
                    assert rel(r.sum, oracle.moments_idx(rows, idx).sum) <= SUM_TOL
                else:
                    assert r.value == 0.0 and r.n == 0
                if q.method != nat.M_EXACT:
                    got = eng.gather(q)
                    assert np.array_equal(got["id"] - 1, idx.astype(np.int64))


def test_numerical_stability_large_offset(nat, oracle):
    """amounts = 1e9 + tiny noise: the shifted/Welford accumulation keeps the variance (hence the CI)
    where a raw sum-of-squares would lose every digit."""
    from approximatequeryengine_amd.engine import Engine, make_query
    n = 200_000
    rows = oracle.synth(n, 7)
    rows["amount"] = 1e9 + (rows["amount"] - 500.0) * 1e-3
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=False)
        idx = oracle.idx_memory_stride(n, 10.0)
        m = oracle.moments_idx(rows, idx)
        r = eng.reduce(make_query(nat.M_MEMORY_STRIDE, 10.0, agg=nat.AVG))
        assert rel(r.value, m.sum / m.n) <= 1e-12
        assert rel(r.m2, m.m2) <= 1e-6
        moe, lo, hi = oracle.ci_cli(nat.AVG, n, m.n, m.m2, m.sum / m.n)
        assert rel(r.ci_upper - r.ci_lower, hi - lo) <= 1e-6


def test_persistent_sweep_stress_alternating_queries(nat, oracle, table, engines):
    """Back-to-back persistent launches with different round counts, early stops and full sweeps: every
    result must equal its first execution (tickets return to zero, epochs separate launches)."""
    from approximatequeryengine_amd.engine import make_query
    eng = engines(1_000_000)
    qs = [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=e, clt_round0=r0, clt_growth=g, num_threads=T)
          for e, r0, g, T in ((1.0, 4096, 4, 4), (0.0, 4096, 4, 4), (0.3, 16, 2, 8), (5.0, 64, 2, 4), (0.05, 1024, 2, 6),
                              (2.0, 8, 2, 2), (0.0, 100_000, 2, 4))]
    for flags in (0, nat.Q_NO_LEAN):  # the lean single launch (default where the sweep is small), and the monitor-wave kernel
        for q in qs:
            q.flags = flags
        first = [eng.reduce(q) for q in qs]
        for rep in range(15):
            for q, f in zip(qs, first):
                r = eng.reduce(q)
                assert r.device_status == 0
                assert (r.n, r.sum, r.sumsq, r.converged, r.rounds, r.topup, r.value) == \
                    (f.n, f.sum, f.sumsq, f.converged, f.rounds, f.topup, f.value), (rep, q.max_error_percent)
    # a non-CLT query in between uses the ordinary launch and must not disturb anything
    assert eng.reduce(make_query(nat.M_BLOCK, 5.0)).n == 50_000
    assert eng.reduce(qs[0]).sum == first[0].sum


def test_single_launch_kernel_choice(nat, engines):
    """Which kernel sweeps a CLT plan's rounds (aqe_plan_last_kernel): the lean launch where the whole sweep is in flight at
    once and every family is a plain run of rows (the bench query through its stride-major views), the persistent sweep
    with its monitor wave when asked for (AQE_Q_NO_LEAN), when the column is swept in place (strided pointers: no runs) and
    when the sweep is long enough for stopping early to matter; one k_round per round under AQE_Q_NO_PERSIST."""
    import torch
    from approximatequeryengine_amd.engine import make_query
    eng = engines(10_000_000)
    st = torch.cuda.Stream().cuda_stream
    def kernel_of(flags, **kw):
        args = dict(max_error_percent=0.01, clt_round0=4096, clt_growth=4)
        args.update(kw)
        q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, **args)
        q.flags = flags
        p = eng.plan(q)
        p.enqueue_all(st)
        r = p.fetch(st)
        k = p.last_kernel()
        p.close()
        return k, r
    k_lean, r_lean = kernel_of(0)
    k_mon, r_mon = kernel_of(nat.Q_NO_LEAN)
    k_place, r_place = kernel_of(nat.Q_NO_LAYOUT)
    k_rounds, r_rounds = kernel_of(nat.Q_NO_PERSIST)
    assert (k_lean, k_mon, k_place, k_rounds) == (nat.KERNEL_SWEEP_LEAN, nat.KERNEL_SWEEP_PERSIST, nat.KERNEL_SWEEP_PERSIST, nat.KERNEL_ROUND)
    for r in (r_mon, r_place, r_rounds):
        assert (r.n, r.visited, r.converged, r.rounds, r.topup) == (r_lean.n, r_lean.visited, r_lean.converged, r_lean.rounds, r_lean.topup)
        assert rel(r.sum, r_lean.sum) <= 1e-14 and rel(r.sumsq, r_lean.sumsq) <= 1e-14 and rel(r.ci_lower, r_lean.ci_lower) <= 1e-13
    # 32 pointers: 96 runs, two per lane of the run table
    k_wide, r_wide = kernel_of(0, num_threads=32)
    k_wide_mon, r_wide_mon = kernel_of(nat.Q_NO_LEAN, num_threads=32)
    assert (k_wide, k_wide_mon) == (nat.KERNEL_SWEEP_LEAN, nat.KERNEL_SWEEP_PERSIST)
    assert (r_wide.n, r_wide.visited, r_wide.converged, r_wide.rounds) == (r_wide_mon.n, r_wide_mon.visited, r_wide_mon.converged, r_wide_mon.rounds)
    assert rel(r_wide.sum, r_wide_mon.sum) <= 1e-14 and rel(r_wide.sumsq, r_wide_mon.sumsq) <= 1e-14 and rel(r_wide.ci_lower, r_wide_mon.ci_lower) <= 1e-13
    # 64 and 256 pointers: 192 and 512 runs — the wide form of the lean launch (run table in LDS, found by bisection)
    for T in (64, 256):
        k_w, r_w = kernel_of(0, num_threads=T)
        k_m, r_m = kernel_of(nat.Q_NO_LEAN, num_threads=T)
        assert (k_w, k_m) == (nat.KERNEL_SWEEP_LEAN, nat.KERNEL_SWEEP_PERSIST), T
        assert (r_w.n, r_w.visited, r_w.converged, r_w.rounds) == (r_m.n, r_m.visited, r_m.converged, r_m.rounds)
        assert rel(r_w.sum, r_m.sum) <= 1e-14 and rel(r_w.sumsq, r_m.sumsq) <= 1e-14 and rel(r_w.ci_lower, r_m.ci_lower) <= 1e-13
    # a batch of early-stopping queries: lean groups, each its plan's head form with the top-up as a slot
    from approximatequeryengine_amd.engine import Batch
    qs = [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=(nat.AVG, nat.SUM)[i % 2], max_error_percent=1.0 + 0.01 * i, clt_round0=4096, clt_growth=4,
                     num_threads=4 + 2 * (i % 3)) for i in range(6)]
    plans = [eng.plan(q) for q in qs]
    alone = []
    for p in plans:
        p.enqueue_all(st)
        alone.append(p.fetch(st))
    b = Batch(plans)
    for _ in range(3):
        b.enqueue_all(st)
        got = b.fetch()
        assert plans[0].last_kernel() == nat.KERNEL_SWEEP_LEAN_MULTI
        for r, w in zip(got, alone):
            assert w.converged == 1 and w.topup > 0
            assert (r.n, r.visited, r.converged, r.rounds, r.topup, r.topup_pending) == (w.n, w.visited, w.converged, w.rounds, w.topup, 0)
            assert rel(r.sum, w.sum) <= 1e-13 and rel(r.ci_lower, w.ci_lower) <= 1e-12
    b.close()
    for p in plans:
        p.close()
    # the head form of a query predicted to stop early is lean as well
    k_head, r_head = kernel_of(0, max_error_percent=1.0)
    k_head_mon, r_head_mon = kernel_of(nat.Q_NO_LEAN, max_error_percent=1.0)
    assert (k_head, k_head_mon) == (nat.KERNEL_SWEEP_LEAN, nat.KERNEL_SWEEP_PERSIST)
    assert (r_head.n, r_head.converged, r_head.rounds, r_head.topup) == (r_head_mon.n, r_head_mon.converged, r_head_mon.rounds, r_head_mon.topup)
    assert rel(r_head.sum, r_head_mon.sum) <= 1e-14 and rel(r_head.ci_upper, r_head_mon.ci_upper) <= 1e-13


def test_concurrent_plans_on_separate_streams(nat, engines):
    """Several plans in flight at once (what bench.py does): each plan owns its hand-off scratch, so queries on
    different HIP streams may overlap — unevenly loaded, back to back — and every one must still return exactly
    what it returns alone."""
    import torch
    from approximatequeryengine_amd.engine import make_query
    eng = engines(1_000_000)
    qs = [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.0, clt_round0=4096, clt_growth=4),
          make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=64, clt_growth=2),
          make_query(nat.M_EXACT, 100.0),
          make_query(nat.M_BLOCK, 20.0, where=(250.0, 750.0)),
          make_query(nat.M_CLT_DUAL_POINTER, 10.0, agg=nat.SUM, max_error_percent=0.2, clt_round0=16, clt_growth=2, num_threads=8),
          make_query(nat.M_MEMORY_STRIDE, 1.0)]
    alone = [eng.reduce(q) for q in qs]
    plans = [eng.plan(q) for q in qs]
    streams = [torch.cuda.Stream() for _ in qs]
    for rep in range(25):
        order = list(range(len(qs)))
        if rep % 2:
            order.reverse()
        for i in order:
            for _ in range(1 + (i + rep) % 3):  # uneven: some streams get several executions per turn
                plans[i].enqueue_all(streams[i].cuda_stream)
        for i, (p, a) in enumerate(zip(plans, alone)):
            r = p.fetch(streams[i].cuda_stream)
            assert r.device_status == 0
            assert (r.n, r.visited, r.sum, r.sumsq, r.value, r.ci_lower, r.converged, r.rounds, r.topup) == \
                (a.n, a.visited, a.sum, a.sumsq, a.value, a.ci_lower, a.converged, a.rounds, a.topup), (rep, i)
    for p in plans:
        p.close()


def test_run_to_run_bitwise_reproducible(nat, engines):
    from approximatequeryengine_amd.engine import make_query
    eng = engines(1_000_000)
    for q in (make_query(nat.M_MEMORY_STRIDE, 20.0), make_query(nat.M_BLOCK, 5.0),
              make_query(nat.M_CLT_DUAL_POINTER, 20.0, max_error_percent=0.0, clt_round0=2048, clt_growth=2)):
        a = eng.reduce(q)
        for _ in range(20):
            b = eng.reduce(q)
            assert (a.sum, a.sumsq, a.value, a.ci_lower, a.n) == (b.sum, b.sumsq, b.value, b.ci_lower, b.n)


def test_device_side_seeded_sampler(nat, oracle, golden, table, engines):
    """AQE_M_RANDOM_DEVICE: a simple random sample without replacement drawn in the kernel (keyed bijection of the rows,
    no host index list).  Bit-exact against its numpy restatement (tests/helpers.perm_rows): the gathered rows are
    exactly P_seed(0..target-1) in draw order, sums match the oracle's moments over those rows, WHERE and row windows
    compose, shards partition the sample.  Statistically it must behave like the reference's own random samplers:
    the recorded runs of parallel_sum_sample (random_device-seeded, DB.cpp:276-304) in the golden file."""
    from approximatequeryengine_amd.engine import Engine, make_query
    from helpers import perm_rows
    for n in (10_000, 100_007, 1_000_000):
        rows, eng = table(n), engines(n)
        for pct, seed in ((1.0, 42), (7.5, 0), (100.0, 123456789), (0.001, 5)):
            want = perm_rows(n, pct, seed)
            q = make_query(nat.M_RANDOM_DEVICE, pct, seed=seed)
            res = eng.reduce(q)
            assert res.visited == len(want) == res.n, (n, pct, seed, res.visited, res.n, len(want), res.device_status, res.rounds)
            got = eng.gather(q)
            assert np.array_equal((got["id"] - 1).astype(np.uint64), want)  # draw order
            if len(want):
                m = oracle.moments_idx(rows, np.sort(want))
                assert rel(res.sum, m.sum) <= SUM_TOL and rel(res.sumsq, m.sumsq) <= SUM_TOL
                assert rel(res.value, m.sum * n / m.n) <= EST_TOL
                rw = eng.reduce(make_query(nat.M_RANDOM_DEVICE, pct, seed=seed, where=(250.0, 750.0), convention=nat.EST_CPP))
                mw = oracle.moments_idx(rows, np.sort(want), where=(250.0, 750.0))
                assert rw.n == mw.n and rw.visited == len(want) and (mw.n == 0 or rel(rw.sum, mw.sum) <= SUM_TOL)
        # a row window is the table (key-range pruning)
        lo, hi = n // 3, n // 3 + n // 2
        r = eng.reduce(make_query(nat.M_RANDOM_DEVICE, 2.0, seed=9, rows=(lo, hi)))
        want = perm_rows(hi - lo, 2.0, 9) + np.uint64(lo)
        m = oracle.moments_idx(rows, np.sort(want))
        assert r.visited == len(want) and rel(r.sum, m.sum) <= SUM_TOL
    # shards: every shard keeps the rows it holds — the union is the single-device sample
    n = 1_000_003
    rows = oracle.synth(n, 42)
    want = perm_rows(n, 1.0, 77)
    tot_n, tot_s = 0, 0.0
    for g, G in ((0, 3), (1, 3), (2, 3)):
        lo, hi = (g * n) // G, ((g + 1) * n) // G
        with Engine(0) as e:
            e.stage_records(rows[lo:hi], shard_lo=lo, n_global=n, keep_aos=False)
            import torch
            vec = torch.zeros(8, dtype=torch.float64, device="cuda")
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                p = e.plan(make_query(nat.M_RANDOM_DEVICE, 1.0, seed=77))
                assert p.rounds == 1
                p.enqueue_round(0, vec.data_ptr(), side.cuda_stream)
                side.synchronize()
                v = vec.cpu().numpy()
                p.close()
            inside = want[(want >= lo) & (want < hi)]
            assert int(v[0]) == len(inside) == int(v[6])
            tot_n += int(v[0])
            tot_s += float(v[1]) + e.info().shift * float(v[0])
    assert tot_n == len(want) and rel(tot_s, float(np.sum(rows["amount"][want.astype(np.int64)]))) <= 1e-12
    # statistical parity with the reference's random_device-seeded reducer (30 recorded runs, N = 1 M, pct 1)
    d = golden["tables"]["1000000"]["distributions"]
    exact = golden["tables"]["1000000"]["exact"]["sum_amount"]
    ref = np.array(d["parallel_sum_pct1_T4"])
    eng = engines(1_000_000)
    ours = np.array([eng.reduce(make_query(nat.M_RANDOM_DEVICE, 1.0, seed=s_, convention=nat.EST_CPP)).value for s_ in range(30)])
    sigma = 288.4 * 100 * math.sqrt(10_000)
    assert np.all(np.abs(ours - exact) <= 4 * sigma) and np.all(np.abs(ref - exact) <= 4 * sigma)
    assert abs(ours.mean() - exact) <= 4 * sigma / math.sqrt(30) and 0.5 * ref.std() <= ours.std() <= 2.0 * ref.std()


def test_stride_view_lifecycle(nat, oracle, table):
    """Stride-major views of the column (one per pointer step in use): at most 8 per table; a ninth step evicts the
    least recently used view that only the reduce cache's plans hold; when live plans hold all eight, the new step is
    swept in place (counted, same answer); restaging drops every view and invalidates the plans laid out over them."""
    from approximatequeryengine_amd.engine import Engine, make_query
    n = 1_000_000
    rows = table(n)

    def q_of(step):
        return make_query(nat.M_MEMORY_STRIDE, 100.0 / step + 1e-9, stride_bytes=32 * step)

    def check(eng, step, res):
        idx = oracle.idx_memory_stride(n, 100.0 / step + 1e-9, 32 * step)
        m = oracle.moments_idx(rows, idx)
        assert res.visited == len(idx) and res.n == m.n and rel(res.sum, m.sum) <= SUM_TOL and rel(res.sumsq, m.sumsq) <= SUM_TOL

    import os
    with Engine(0) as eng:  # the default budget (16 GiB of views): a small table simply gets a view per step, nothing is evicted
        eng.stage_records(rows, keep_aos=False)
        for step in range(2, 13):
            check(eng, step, eng.reduce(q_of(step)))
        t = eng.info()
        assert (t.n_views, t.view_evictions, t.view_fallbacks) == (11, 0, 0)
    os.environ["AQE_VIEW_BUDGET_MB"] = "0"  # from here on: the eight views a table may always hold, and no more
    try:
        _stride_view_lifecycle_at_the_cap(nat, eng_rows=(rows, n), q_of=q_of, check=check)
    finally:
        del os.environ["AQE_VIEW_BUDGET_MB"]


def _stride_view_lifecycle_at_the_cap(nat, eng_rows, q_of, check):
    from approximatequeryengine_amd.engine import Engine, make_query
    rows, n = eng_rows
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=False)
        base = eng.info()
        assert (base.n_views, base.view_bytes, base.view_evictions, base.view_fallbacks) == (0, 0, 0, 0)
        for step in range(2, 10):  # eight steps, eight views (held by the reduce cache's plans only)
            check(eng, step, eng.reduce(q_of(step)))
        t = eng.info()
        assert t.n_views == 8 and t.view_evictions == 0 and t.view_bytes >= 8 * 8 * n and t.hbm_bytes == base.hbm_bytes + t.view_bytes
        check(eng, 2, eng.reduce(q_of(2)))  # touch step 2: step 3 is now the least recently used
        check(eng, 10, eng.reduce(q_of(10)))  # the ninth step: one view goes
        t = eng.info()
        assert (t.n_views, t.view_evictions, t.view_fallbacks) == (8, 1, 0)
        check(eng, 3, eng.reduce(q_of(3)))  # the evicted step comes back (and evicts another): same answer
        assert eng.info().view_evictions == 2
        # live plans hold their views: with all eight held, a new step is laid out in place
        plans = []
        for step in (2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12):
            p = eng.plan(q_of(step))
            p.enqueue_all(0)
            check(eng, step, p.fetch(0))
            plans.append(p)
        t = eng.info()
        assert t.n_views == 8 and t.view_fallbacks >= 1, (t.n_views, t.view_fallbacks)
        for p in plans[:4]:
            p.close()
        before = eng.info().view_evictions
        check(eng, 13, eng.reduce(q_of(13)))  # released views can be evicted again
        assert eng.info().view_evictions == before + 1
        # restaging: views gone, plans over the old table refuse to run
        eng.stage_records(rows[: n // 2], keep_aos=False)
        t = eng.info()
        assert (t.n_views, t.view_bytes, t.view_evictions, t.view_fallbacks) == (0, 0, 0, 0) and t.global_rows == n // 2
        with pytest.raises(nat.AqeError):
            plans[-1].enqueue_all(0)
        for p in plans[4:]:
            p.close()
        r = eng.reduce(make_query(nat.M_MEMORY_STRIDE, 20.0))
        assert r.visited == (n // 2) // 5 and eng.info().n_views == 1


def test_file_staging_round_trip(nat, oracle, golden, table, tmp_path):
    """The reference's file format (DB.cpp:665-711): stage from a file, save, byte-identical."""
    from approximatequeryengine_amd.engine import Engine, make_query
    n = 50_000
    rows = table(100_000)[:n]
    p = tmp_path / "t.db"
    assert oracle.file_write(p, rows) == 0
    with Engine(0) as eng:
        eng.stage_file(p, keep_aos=True)
        assert eng.info().global_rows == n
        r = eng.reduce(make_query(nat.M_EXACT, 100.0))
        assert rel(r.value, math.fsum(rows["amount"])) <= SUM_TOL
        q = tmp_path / "out.db"
        eng.save_file(q)
        assert q.read_bytes()[24:] == p.read_bytes()[24:]
        assert q.read_bytes()[:8] == p.read_bytes()[:8] and q.read_bytes()[16:24] == p.read_bytes()[16:24]
        # a shard of the file, staged amount-only
        eng.stage_file(p, shard_lo=10_000, n_local=20_000, keep_aos=False)
        t = eng.info()
        assert (t.global_rows, t.shard_lo, t.local_rows, t.has_aos) == (n, 10_000, 20_000, 0)
        assert t.shift == float(np.sum(rows["amount"][:1024], dtype=np.float64) / 1024) or abs(t.shift - rows["amount"][:1024].mean()) < 1e-9
        # hostile / truncated headers are I/O errors, not reads past the mapping: a row count that wraps 24 + 32 * count
        # around 2^64 (2^59 + 1 rows in a 56-byte file), a count one row beyond the file, a file shorter than its header
        import struct
        for name, blob in (("wrap.db", struct.pack("<QQQ", 1, 1, (1 << 59) + 1) + b"\0" * 32),
                           ("short.db", p.read_bytes()[: 24 + 32 * 100 - 1]),
                           ("tiny.db", b"\0" * 23)):
            bad = tmp_path / name
            bad.write_bytes(blob if name != "short.db" else struct.pack("<QQQ", 100, 1, 100) + blob[24:])
            with pytest.raises(nat.AqeError) as ei:
                eng.stage_file(bad, keep_aos=False)
            assert ei.value.status == nat.ERR_IO
            with pytest.raises(nat.AqeError):
                eng.stage_file(bad, shard_lo=0, n_local=1, keep_aos=True)


def test_file_staging_several_pinned_chunks(nat, oracle, tmp_path):
    """Config 3's staging path at a size that needs several turns of the pinned ring (four buffers of 256 Ki rows = 8 MiB
    each for a table this size): a 5 M-row file in the reference's format, pread -> pinned ring -> HBM, whole and as the
    middle shard of three; the exact SUM, a strided sample and the rows themselves against the host copy; and where the
    time went (aqe_last_stage_stats): the ring is pinned by the context's first staging only."""
    from approximatequeryengine_amd.engine import Engine, make_query
    n = 5_000_000
    rows = oracle.synth(n, 7)
    p = tmp_path / "big.db"
    assert oracle.file_write(p, rows) == 0
    with Engine(0) as eng:
        eng.stage_file(p, keep_aos=True)
        assert eng.info().global_rows == n and eng.info().has_aos == 1
        st = eng.stage_stats()
        assert st.chunks == (n + (1 << 18) - 1) >> 18 and st.host_bytes == 32 * n == st.link_bytes and st.fill_threads >= 1
        assert st.pinned_alloc_ms > 0.0 and st.total_ms >= st.fill_ms > 0.0
        r = eng.reduce(make_query(nat.M_EXACT, 100.0))
        assert r.n == n and rel(r.value, math.fsum(rows["amount"])) <= SUM_TOL
        got = eng.gather(make_query(nat.M_MEMORY_STRIDE, 1.0))
        assert got.tobytes() == rows[::100][: len(got)].tobytes() and len(got) == n // 100
        lo, hi = n // 3, 2 * n // 3
        eng.stage_file(p, shard_lo=lo, n_local=hi - lo, keep_aos=False)  # amounts only: a quarter of the bytes cross PCIe
        st = eng.stage_stats()
        assert st.pinned_alloc_ms == 0.0 and st.host_bytes == 32 * (hi - lo) and st.link_bytes == 8 * (hi - lo)  # (the ring was there)
        t = eng.info()
        assert (t.global_rows, t.shard_lo, t.local_rows, t.has_aos) == (n, lo, hi - lo, 0)
        import torch
        vec = torch.zeros(8, dtype=torch.float64, device="cuda")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            pl = eng.plan(make_query(nat.M_EXACT, 100.0))
            pl.enqueue_round(0, vec.data_ptr(), side.cuda_stream)
            side.synchronize()
            v = vec.cpu().numpy()
            pl.close()
        assert int(v[0]) == hi - lo and rel(float(v[1]) + t.shift * float(v[0]), math.fsum(rows["amount"][lo:hi])) <= SUM_TOL


def test_synthetic_generator_matches_oracle(nat, oracle):
    from approximatequeryengine_amd.engine import Engine, make_query
    n = 300_001
    rows = oracle.synth(n, 42)
    with Engine(0) as eng:
        eng.generate_synthetic(n, seed=42, keep_aos=True)
        assert abs(eng.info().shift - rows["amount"][:1024].mean()) < 1e-9
        got = eng.gather(make_query(nat.M_MEMORY_STRIDE, 100.0))
        assert got.tobytes() == rows.tobytes()
        r = eng.reduce(make_query(nat.M_EXACT, 100.0))
        assert rel(r.value, math.fsum(rows["amount"])) <= SUM_TOL


@pytest.mark.parametrize("n", [100_007, 1_000_003], ids=["100k", "1M"])
def test_sharded_plan_api_single_gpu_virtual_shards(nat, oracle, table, n):
    """G virtual shards on one GPU through the stepwise API: per-round partial vectors summed on the host
    stand in for the all-reduce; result identical to the single-shard query for G in {1,2,3,8}.
    (At 1 M rows the CLT sweeps are large enough to read the shards' stride-major views of the column.)"""
    import ctypes as C
    from approximatequeryengine_amd.engine import Engine, make_query
    import torch
    rows = table(n)
    queries = [
        make_query(nat.M_MEMORY_STRIDE, 1.0),
        make_query(nat.M_BLOCK, 5.0, where=(250.0, 750.0)),
        make_query(nat.M_RANDOM_POINTER, 2.0, seed=9),
        make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=256, clt_growth=2, num_threads=8),
        make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.0, clt_round0=4096, clt_growth=2),
    ]
    with Engine(0) as whole:
        whole.stage_records(rows, keep_aos=False)
        refs = [whole.reduce(q) for q in queries]
    for G in (1, 2, 3, 8):
        bounds = [(g * n) // G for g in range(G + 1)]
        engs = []
        for g in range(G):
            e = Engine(0)
            e.stage_records(rows[bounds[g]:bounds[g + 1]], shard_lo=bounds[g], n_global=n, keep_aos=False)
            e.set_shift(float(rows["amount"][:1024].mean()))
            engs.append(e)
        side = torch.cuda.Stream()
        st = side.cuda_stream
        for q, want in zip(queries, refs):
            plans = [e.plan(q) for e in engs]
            rounds, topup = plans[0].rounds, plans[0].has_topup
            with torch.cuda.stream(side):
                vecs = torch.zeros(G, nat.MOMENT_VEC, dtype=torch.float64, device="cuda")
                for p in plans:
                    p.reset(st)
                for r in range(rounds + (1 if topup else 0)):
                    vecs.zero_()  # a launch that leaves early (should_stop) writes nothing
                    for g, p in enumerate(plans):
                        p.enqueue_round(r, vecs[g].data_ptr(), st)
                    total = vecs.sum(0)  # stand-in for the all-reduce, same stream => ordered
                    for p in plans:
                        p.enqueue_update(r, total.data_ptr(), st)
                outs = []
                for p in plans:
                    p.enqueue_finalize(st)
                    outs.append(p.fetch(st))
            for o_ in outs:
                assert (o_.n, o_.visited, o_.converged, o_.rounds, o_.topup) == (want.n, want.visited, want.converged, want.rounds, want.topup)
                assert rel(o_.sum, want.sum) <= SUM_TOL and rel(o_.value, want.value) <= EST_TOL
                assert rel(o_.ci_lower, want.ci_lower) <= EST_TOL
            # batched form: every round swept in one launch per shard, ONE reduction, decisions replayed; a due
            # top-up comes back as a mark and is run as the stepwise top-up step
            if plans[0].totals_len:
                assert all(p.totals_len == plans[0].totals_len == rounds * nat.MOMENT_VEC for p in plans)
                with torch.cuda.stream(side):
                    tot = torch.full((G, plans[0].totals_len), float("nan"), dtype=torch.float64, device="cuda")
                    for g, p in enumerate(plans):
                        p.enqueue_sweep_totals(tot[g].data_ptr(), st)
                    red = tot.sum(0)
                    outs = []
                    for p in plans:
                        p.enqueue_replay(red.data_ptr(), st)
                        outs.append(p.fetch(st))
                    marks = {o_.topup_pending for o_ in outs}
                    assert len(marks) == 1 and marks <= {0, 1} and (marks == {0} or topup)
                    assert marks == {1 if want.topup else 0}
                    if marks == {1}:
                        vecs.zero_()
                        for g, p in enumerate(plans):
                            p.enqueue_round(rounds, vecs[g].data_ptr(), st)
                        total = vecs.sum(0)
                        outs = []
                        for p in plans:
                            p.enqueue_update(rounds, total.data_ptr(), st)
                            p.enqueue_finalize(st)
                            outs.append(p.fetch(st))
                for o_ in outs:
                    assert (o_.n, o_.visited, o_.converged, o_.rounds, o_.topup) == (want.n, want.visited, want.converged, want.rounds, want.topup)
                    assert rel(o_.sum, want.sum) <= SUM_TOL and rel(o_.value, want.value) <= EST_TOL
                    assert rel(o_.ci_lower, want.ci_lower) <= EST_TOL and o_.device_status == 0
            else:
                assert q.method != nat.M_CLT_DUAL_POINTER
            for p in plans:
                p.close()
        for e in engs:
            e.close()


@pytest.mark.parametrize("n", [100_000_000, 1_000_000_000], ids=["100M", "1B"])
def test_full_size_properties(nat, oracle, n):
    """BASELINE.json's full table sizes (configs 3-5: 100 M and 1 B rows), where the oracle cannot be run:
    size-independent properties instead — additivity of the reducer over row windows and over a WHERE partition,
    the reference's sample counts, the persistent sweep against one launch per round, and the synthetic data's
    known moments.  (The monitor works through several windows of its partial list only on tables this large.)"""
    from approximatequeryengine_amd.engine import Engine, make_query
    with Engine(0) as eng:
        eng.generate_synthetic(n, keep_aos=False)
        whole = eng.reduce(make_query(nat.M_EXACT, 100.0))
        assert whole.n == whole.visited == n
        # uniform [1, 1000): mean 500.5, sigma 288.39 -> the exact mean is within 6 sigma / sqrt(n)
        assert abs(whole.sum / n - 500.5) < 6 * 288.39 / n ** 0.5
        # additivity over row windows (ragged cuts) ...
        cuts = [0, 1, 4097, n // 7, n // 3 + 5, n // 2, n - 100_003, n - 1, n]
        parts = [eng.reduce(make_query(nat.M_EXACT, 100.0, rows=(a, b))) for a, b in zip(cuts[:-1], cuts[1:])]
        assert sum(p.n for p in parts) == n
        assert rel(math.fsum(p.sum for p in parts), whole.sum) <= 1e-12
        assert rel(math.fsum(p.sumsq for p in parts), whole.sumsq) <= 1e-12
        # ... and over a partition of the value range (both ends inclusive, DB.cpp:329)
        lo_hi = [(1.0, 250.0), (float(np.nextafter(250.0, 1e9)), 750.0), (float(np.nextafter(750.0, 1e9)), 1000.0)]
        bands = [eng.reduce(make_query(nat.M_EXACT, 100.0, where=w)) for w in lo_hi]
        assert sum(b.n for b in bands) == n and all(b.visited == n for b in bands)
        assert rel(math.fsum(b.sum for b in bands), whole.sum) <= 1e-12
        # the reference's sample counts (index generators run count-only)
        for q, want in (
            (make_query(nat.M_MEMORY_STRIDE, 1.0), oracle.count("aqo_idx_memory_stride", n, 1.0, 0)),
            (make_query(nat.M_BLOCK, 1.0, where=(250.0, 750.0)), oracle.count("aqo_idx_block", n, 1.0, 1000)),
            (make_query(nat.M_BLOCK, 20.0), oracle.count("aqo_idx_block", n, 20.0, 1000)),
            (make_query(nat.M_PAGE, 5.0, block_size=4096), oracle.count("aqo_idx_page", n, 5.0, 4096)),
            (make_query(nat.M_OPTIMIZED_CLT, 20.0), oracle.count("aqo_idx_optimized_clt", n, 20.0, 4)),
        ):
            r = eng.reduce(q)
            assert r.visited == want and (r.n == want or q.has_where)
            # a sample mean of uniform data: within 6 standard errors of the table's mean (WHERE band: of 500)
            centre, sigma = (500.0, 144.4) if q.has_where else (whole.sum / n, 288.39)
            assert abs(r.sum / r.n - centre) < 6 * sigma / r.n ** 0.5 + 1e-9
        # GROUP BY: the bins partition the sample — counts add up, sums add up to the ungrouped query's
        for q in (make_query(nat.M_EXACT, 100.0), make_query(nat.M_BLOCK, 20.0, where=(250.0, 750.0)), make_query(nat.M_ROWID_MOD, 3.0)):
            flat = eng.reduce(q)
            for column, groups in ((nat.GROUP_REGION, 4), (nat.GROUP_PRODUCT, 100)):
                got = eng.reduce_grouped(q, column)
                assert len(got) == groups and [g_.key for g_ in got] == list(range(groups))
                assert sum(g_.n for g_ in got) == flat.n and sum(g_.visited for g_ in got) == flat.visited
                assert rel(math.fsum(g_.sum for g_ in got), flat.sum) <= 1e-12
                assert rel(math.fsum(g_.sumsq for g_ in got), flat.sumsq) <= 1e-12
        # CLT monitor: one persistent launch == one launch per round, for a sweep that never converges, one that
        # stops in the middle and one that stops at once
        for e in (0.0, 0.005, 0.05, 1.0):
            q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=e, clt_round0=4096, clt_growth=4)
            q.flags = nat.Q_NO_PERSIST
            multi = eng.reduce(q)
            q.flags = nat.Q_FORCE_PERSIST
            res = eng.reduce(q)
            assert (res.n, res.visited, res.converged, res.rounds, res.topup) == (multi.n, multi.visited, multi.converged, multi.rounds, multi.topup)
            assert rel(res.sum, multi.sum) <= 1e-13 and rel(res.sumsq, multi.sumsq) <= 1e-13 and rel(res.ci_lower, multi.ci_lower) <= 1e-12
            if e == 0.0:  # the reference's full dual-pointer sweep: 2 x base rows, no top-up
                assert res.n == 2 * int(n * 20.0 / 100.0) and res.topup == 0 and res.converged == 0
            # ... and == the batched multi-GPU form on a world of one shard (every round's total in one launch,
            # replayed; the top-up, when due, as the stepwise step)
            import torch
            plan = eng.plan(q)
            tot = torch.full((plan.totals_len,), float("nan"), dtype=torch.float64, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            plan.enqueue_sweep_totals(tot.data_ptr(), st)
            plan.enqueue_replay(tot.data_ptr(), st)
            b = plan.fetch(st)
            assert b.topup_pending == (1 if res.topup else 0) and (b.converged, b.rounds) == (res.converged, res.rounds)
            if b.topup_pending:
                vec = torch.zeros(nat.MOMENT_VEC, dtype=torch.float64, device="cuda")
                plan.enqueue_round(plan.rounds, vec.data_ptr(), st)
                plan.enqueue_update(plan.rounds, vec.data_ptr(), st)
                plan.enqueue_finalize(st)
                b = plan.fetch(st)
            assert (b.n, b.visited, b.topup) == (res.n, res.visited, res.topup) and rel(b.sum, res.sum) <= 1e-13 and rel(b.ci_upper, res.ci_upper) <= 1e-12
            plan.close()
            assert abs(res.value - whole.sum / n) <= 3.0 * (res.ci_upper - res.ci_lower) / 2  # 95 % half-width x 3 ~ 6 sigma


def test_native_batch_drives_several_plans_with_two_calls_per_step(nat, table):
    """aqe_batch: sweeps of several plans on the library's side streams, one buffer for the collective, replays
    back on the side streams; same answers as the single-GPU path, step after step."""
    from approximatequeryengine_amd.distributed import ShardedBatch
    from approximatequeryengine_amd.engine import Batch, Engine, make_query
    import torch
    n = 1_000_000
    rows = table(n)
    qs = [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=e, clt_round0=r0, clt_growth=g, num_threads=t)
          for e, r0, g, t in ((0.0, 4096, 4, 4), (5.0, 256, 2, 8), (0.3, 16, 2, 8), (0.0, 4096, 4, 4))]  # (e = 5 %: stops short of base/4 -> top-up)
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=False)
        want = [eng.reduce(q) for q in qs]
        plans = [eng.plan(q) for q in qs]
        width = max(p.totals_len for p in plans) + 8  # a row stride wider than any plan needs
        side = torch.cuda.Stream()
        calls = [0]

        def all_reduce(t):  # a world of one: the identity, but it must see the sweeps' output
            calls[0] += 1
            assert bool(torch.isfinite(t[:, : plans[0].totals_len][0]).all())

        with torch.cuda.stream(side):
            buf = torch.full((len(plans), width), float("nan"), dtype=torch.float64, device="cuda")
            native = Batch(plans)
            sb = ShardedBatch(plans, buf, all_reduce, stream=side.cuda_stream, batch=native)
            for _ in range(3):
                for _ in range(4):
                    sb.enqueue()          # steps back to back, no host synchronisation in between
                got = sb.fetch()
                assert calls[0] % 4 == 0
                assert [g_.topup_pending for g_ in got] == [1 if w.topup else 0 for w in want]
                for g_, w in zip(got, want):
                    assert (g_.converged, g_.rounds, g_.device_status) == (w.converged, w.rounds, 0)
                    if not w.topup:
                        assert (g_.n, g_.visited) == (w.n, w.visited)
                        assert rel(g_.sum, w.sum) <= 1e-13 and rel(g_.ci_lower, w.ci_lower) <= 1e-12
            out = sb.run()                # run() also finishes the due top-ups with the stepwise step
            for g_, w in zip(out, want):
                assert (g_.n, g_.visited, g_.converged, g_.rounds, g_.topup, g_.topup_pending) == (w.n, w.visited, w.converged, w.rounds, w.topup, 0)
                assert rel(g_.sum, w.sum) <= 1e-13 and rel(g_.value, w.value) <= 1e-13
            assert any(w.topup for w in want) and not all(w.topup for w in want)
            # two batches software-pipelined: a step's collective is issued after the NEXT step's sweeps
            from approximatequeryengine_amd.distributed import PipelinedBatches
            plans2 = [eng.plan(q) for q in qs]
            buf2 = torch.full((len(plans2), width), float("nan"), dtype=torch.float64, device="cuda")
            native2 = Batch(plans2)
            pipe = PipelinedBatches([sb, ShardedBatch(plans2, buf2, all_reduce, stream=side.cuda_stream, batch=native2)])
            before = calls[0]
            for _ in range(7):
                pipe.enqueue()
            got = pipe.fetch()
            assert calls[0] - before == 7 and len(got) == 2 * len(qs)
            for g_, w in zip(got, want + want):
                assert (g_.converged, g_.rounds, g_.device_status, g_.topup_pending) == (w.converged, w.rounds, 0, 1 if w.topup else 0)
                if not w.topup:
                    assert (g_.n, g_.visited) == (w.n, w.visited) and rel(g_.sum, w.sum) <= 1e-13
            native2.close()
            for p in plans2:
                p.close()
            native.close()
        with pytest.raises(nat.AqeError):
            Batch([plans[0], plans[0]])
        for p in plans:
            p.close()


def test_single_round_lean_route_over_row_windows(nat, oracle, table):
    """Key-range windows (`WHERE id BETWEEN`: the sampler runs over the window as if it were the table) through the lean
    single-round route: a window cuts the first and the last block of a block sample wherever it likes, so the segmented
    run comes with plain runs at both ends — and through k_round, the same rows: both against the oracle."""
    from approximatequeryengine_amd.engine import Engine, make_query
    n = 1_234_567
    rows = table(n)
    rng = np.random.default_rng(5)
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=False)
        for _ in range(40):
            lo = int(rng.integers(0, n - 70_000))
            hi = int(rng.integers(lo + 60_000, n + 1))
            w = hi - lo
            kind = int(rng.integers(0, 4))
            if kind == 0:
                q, idx = make_query(nat.M_EXACT, 100.0, rows=(lo, hi)), np.arange(w, dtype=np.uint64)
            elif kind == 1:
                B = int(rng.choice([512, 777, 1000, 1024, 1025, 4096, 5000]))
                pct = float(rng.choice([5.0, 20.0, 33.0, 50.0]))
                q, idx = make_query(nat.M_BLOCK, pct, block_size=B, rows=(lo, hi)), oracle.idx_block(w, pct, B)
            elif kind == 2:
                pct = float(rng.choice([10.0, 20.0, 25.0, 50.0]))
                q, idx = make_query(nat.M_MEMORY_STRIDE, pct, rows=(lo, hi)), oracle.idx_memory_stride(w, pct)
            else:
                B, T, pct = int(rng.choice([600, 1000, 2048])), int(rng.integers(1, 7)), float(rng.choice([10.0, 30.0]))
                q, idx = make_query(nat.M_PARALLEL_BLOCK, pct, block_size=B, num_threads=T, rows=(lo, hi)), oracle.idx_parallel_block(w, pct, B, T)
            where = (150.0, 850.0) if rng.integers(0, 2) else None
            if where:
                q.has_where, q.where_min, q.where_max = 1, where[0], where[1]
            m = oracle.moments_idx(rows, idx + np.uint64(lo), where=where)
            for flags in (0, nat.Q_FORCE_LEAN, nat.Q_NO_LEAN):
                q.flags = flags
                r = eng.reduce(q)
                assert (r.visited, r.n) == (len(idx), m.n), (kind, lo, hi, flags, r.as_dict())
                if m.n:
                    assert rel(r.sum, m.sum) <= SUM_TOL and rel(r.sumsq, m.sumsq) <= SUM_TOL, (kind, lo, hi, flags)


def test_batch_of_single_round_samplers_as_lean_groups(nat, oracle, table):
    """A batch whose plans are single-round samplers made of runs and rows of blocks — exact scans (whole table, a key-range
    window with odd ends), a strided sample through its view, block samples (1000-row blocks: a segmented run; 4096-row
    blocks with WHERE; the window cutting the first and the last block) — runs as lean groups of ONE launch
    (k_sweep_lean_multi): tiles dealt out wave by wave inside a group when it has at least a tile per wave, contiguous
    shares otherwise.  Every query against the oracle's moments over the oracle's index list, step after step."""
    from approximatequeryengine_amd.engine import Batch, Engine, make_query
    n = 3_000_017
    rows = table(n)
    specs = [
        (make_query(nat.M_EXACT, 100.0), np.arange(n, dtype=np.uint64), None),
        (make_query(nat.M_EXACT, 100.0, agg=nat.AVG, rows=(100_003, 2_900_001), where=(100.0, 800.0)), np.arange(100_003, 2_900_001, dtype=np.uint64), (100.0, 800.0)),
        (make_query(nat.M_MEMORY_STRIDE, 20.0), oracle.idx_memory_stride(n, 20.0), None),
        (make_query(nat.M_BLOCK, 20.0), oracle.idx_block(n, 20.0, 1000), None),
        (make_query(nat.M_BLOCK, 10.0, block_size=4096, where=(250.0, 750.0), convention=nat.EST_CPP), oracle.idx_block(n, 10.0, 4096), (250.0, 750.0)),
        (make_query(nat.M_BLOCK, 33.0, block_size=777), oracle.idx_block(n, 33.0, 777), None),
        (make_query(nat.M_PARALLEL_BLOCK, 12.0, block_size=2000, num_threads=4), oracle.idx_parallel_block(n, 12.0, 2000, 4), None),
    ]
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=False)
        plans = [eng.plan(q) for q, _, _ in specs]
        b = Batch(plans)
        for step in range(3):
            b.enqueue_all(0)
            got = b.fetch()
            assert plans[0].last_kernel() == nat.KERNEL_SWEEP_LEAN_MULTI
            for (q, idx, where), r in zip(specs, got):
                m = oracle.moments_idx(rows, idx, where=where)
                assert (r.visited, r.n, r.device_status, r.topup_pending) == (len(idx), m.n, 0, 0), (step, q.method, r.as_dict())
                assert rel(r.sum, m.sum) <= SUM_TOL and rel(r.sumsq, m.sumsq) <= SUM_TOL and rel(r.m2, m.m2) <= 1e-9
        single = [eng.reduce(q) for q, _, _ in specs]  # ... and what each reports as a launch of its own
        for r, w in zip(got, single):
            assert (r.n, r.visited) == (w.n, w.visited) and rel(r.value, w.value) <= 1e-12 and rel(r.ci_lower, w.ci_lower) <= 1e-12
        b.close()
        for p in plans:
            p.close()


def test_batch_in_one_launch_matches_single_plans_and_oracle(nat, oracle, table):
    """aqe_batch_enqueue_all: a mixed batch of queries in ONE launch (a group of workgroups per query: lean groups,
    k_sweep_lean_multi, when every plan's families are plain runs of rows; else k_sweep_multi, a monitor wave and a
    should_stop word per query — the sub-batches below take both).  Every query must report what it reports as a launch of its own and what
    the oracle computes (DB.cpp:885-1043 for the CLT monitor; 1526-1603, 1151-1181, 242-274 for the others) — step
    after step, results fetched every step, including the early-stop / top-up / head-form plans."""
    from approximatequeryengine_amd.engine import Batch, Engine, make_query
    import torch
    n = 1_000_000
    rows = table(n)
    clt = [(0.0, 4096, 4, 4, nat.AVG), (1.0, 256, 2, 8, nat.SUM), (0.3, 16, 2, 8, nat.AVG), (0.01, 4096, 4, 6, nat.COUNT),
           (1.0, 4096, 4, 4, nat.AVG), (0.05, 1024, 4, 5, nat.SUM)]
    qs = [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=agg, max_error_percent=e, clt_round0=r0, clt_growth=g, num_threads=t)
          for e, r0, g, t, agg in clt]
    qs += [make_query(nat.M_MEMORY_STRIDE, 20.0), make_query(nat.M_EXACT, 100.0, where=(250.0, 750.0)),
           make_query(nat.M_BLOCK, 5.0, where=(100.0, 900.0), convention=nat.EST_CPP), make_query(nat.M_PAGE, 5.0, block_size=4096, agg=nat.AVG),
           make_query(nat.M_OPTIMIZED_CLT, 10.0, num_threads=4)]
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=False)
        want = [eng.reduce(q) for q in qs]
        for (e, r0, g, t, agg), w in zip(clt, want):  # the single-plan path itself against the oracle
            rc, o, _ = oracle.clt_run(rows, 20.0, 0.95, 10, t, e, R0=r0, growth=g)
            assert rc == 0 and (w.n, w.converged, w.rounds, w.topup) == (o.final.n, o.converged, o.rounds, o.topup)
            assert rel(w.sum, o.final.sum) <= SUM_TOL
        side = torch.cuda.Stream()
        for sizes in ((len(qs),), (1,), (3, 5)):  # one batch of all; batches of one; two batches in flight on one stream
            for lo in range(0, len(qs), sum(sizes)):
                groups, at = [], lo
                for k in sizes:
                    groups.append(list(range(at, min(at + k, len(qs)))))
                    at += k
                groups = [g_ for g_ in groups if g_]
                plan_sets = [[eng.plan(qs[i]) for i in g_] for g_ in groups]
                batches = [Batch(ps) for ps in plan_sets]
                for step in range(4):
                    for b in batches:
                        b.enqueue_all(side.cuda_stream)
                    for b, g_ in zip(batches, groups):
                        for r, i in zip(b.fetch(), g_):
                            w = want[i]
                            assert (r.n, r.visited, r.converged, r.rounds, r.topup, r.topup_pending, r.device_status) == \
                                   (w.n, w.visited, w.converged, w.rounds, w.topup, 0, 0), (i, step, r.as_dict(), w.as_dict())
                            assert rel(r.sum, w.sum) <= 1e-13 and rel(r.sumsq, w.sumsq) <= 1e-13
                            assert rel(r.value, w.value) <= 1e-12 and rel(r.ci_lower, w.ci_lower) <= 1e-12 and rel(r.ci_upper, w.ci_upper) <= 1e-12
                b0 = batches[0]
                b0.set_profiling(True)
                b0.enqueue_all(side.cuda_stream)
                b0.fetch()
                ms, swept, wgs = b0.launch_info()
                assert ms > 0 and wgs >= 1 and swept > 0
                b0.set_profiling(False)
                for b in batches:
                    b.close()
                for ps in plan_sets:
                    for p in ps:
                        p.close()
        # more queries than compute units: groups of one workgroup each (15 sweeper waves + the monitor), a grid the chip
        # cannot hold at once — later groups wait for earlier ones to leave, every monitor outlasts the wait
        many = [qs[i % len(qs)] for i in range(300)]
        plans = [eng.plan(q) for q in many]
        big = Batch(plans)
        for _ in range(2):
            big.enqueue_all(side.cuda_stream)
            for i, r in enumerate(big.fetch()):
                w = want[i % len(qs)]
                assert (r.n, r.visited, r.converged, r.rounds, r.topup, r.topup_pending, r.device_status) == (w.n, w.visited, w.converged, w.rounds, w.topup, 0, 0), i
                assert rel(r.sum, w.sum) <= 1e-13 and rel(r.ci_lower, w.ci_lower) <= 1e-12
        assert big.launch_info(False)[2] >= 300
        assert plans[0].last_kernel() == nat.KERNEL_SWEEP_MULTI  # (blocks and pages in the batch: the groups with monitor waves)
        big.close()
        for p in plans:
            p.close()
        # the same with plans that all qualify for the lean groups (CLT queries that run to the end or stop in the middle, a
        # strided sample, an exact scan under WHERE; not the early-stopping ones: on a table this small their top-up, 50 000
        # rows, gets no stride-major view and is not a run): 300 groups that wait for nothing
        lean_kinds = [0, 2, 3, 5, 6, 7]
        plans = [eng.plan(qs[lean_kinds[i % len(lean_kinds)]]) for i in range(300)]
        big = Batch(plans)
        for _ in range(2):
            big.enqueue_all(side.cuda_stream)
            for i, r in enumerate(big.fetch()):
                w = want[lean_kinds[i % len(lean_kinds)]]
                assert (r.n, r.visited, r.converged, r.rounds, r.topup, r.topup_pending, r.device_status) == (w.n, w.visited, w.converged, w.rounds, w.topup, 0, 0), i
                assert rel(r.sum, w.sum) <= 1e-13 and rel(r.ci_lower, w.ci_lower) <= 1e-12
        assert big.launch_info(False)[2] >= 300 and plans[0].last_kernel() == nat.KERNEL_SWEEP_LEAN_MULTI
        big.close()
        for p in plans:
            p.close()
        # a seeded-random plan has no single-launch form
        pr = eng.plan(make_query(nat.M_RANDOM_POINTER, 1.0, seed=7))
        br = Batch([pr])
        with pytest.raises(nat.AqeError):
            br.enqueue_all(side.cuda_stream)
        br.close()
        pr.close()


def test_rccl_communicator_behind_the_c_abi(nat, oracle, table):
    """aqe_comm_*: the library's own RCCL communicator (a world of one rank here — the collective really runs through
    librccl), the whole-query and whole-batch host paths in C (aqe_plan_run_sharded, aqe_batch_run_sharded), and the
    Python orchestration driven with its DEFAULT stream argument.  Answers: the single-GPU path's, and the oracle's."""
    from approximatequeryengine_amd.distributed import ShardedBatch, ShardedQuery, native_all_reduce
    from approximatequeryengine_amd.engine import Batch, Comm, Engine, make_query
    import torch
    n = 1_000_000
    rows = table(n)
    qs = [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=e, clt_round0=r0, clt_growth=g, num_threads=t)
          for e, r0, g, t in ((0.0, 4096, 4, 4), (1.0, 256, 2, 8), (0.3, 16, 2, 8))]
    qs += [make_query(nat.M_MEMORY_STRIDE, 1.0), make_query(nat.M_BLOCK, 1.0, where=(250.0, 750.0), convention=nat.EST_CPP),
           make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=2.0)]  # the last: 5000 rounds, no batched form
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=False)
        want = [eng.reduce(q) for q in qs]
        side = torch.cuda.Stream()
        comm = Comm(eng, Comm.unique_id(), 1, 0)
        assert (comm.nranks, comm.rank) == (1, 0)
        with torch.cuda.stream(side):
            t = torch.arange(64, dtype=torch.float64, device="cuda")
            comm.all_reduce_sum(t.data_ptr(), t.numel(), side.cuda_stream)
            comm.all_reduce_max(t.data_ptr(), t.numel(), side.cuda_stream)
            side.synchronize()
            assert torch.equal(t.cpu(), torch.arange(64, dtype=torch.float64))
            vec = torch.zeros(512, dtype=torch.float64, device="cuda")
            for q, w in zip(qs, want):
                p = eng.plan(q)
                r = comm.run_plan(p, vec.data_ptr(), side.cuda_stream)
                assert (r.n, r.visited, r.converged, r.rounds, r.topup, r.topup_pending) == (w.n, w.visited, w.converged, w.rounds, w.topup, 0), q.method
                assert rel(r.sum, w.sum) <= 1e-13 and rel(r.value, w.value) <= 1e-12 and rel(r.ci_lower, w.ci_lower) <= 1e-12
                # the Python orchestration with its defaults (stream=0 -> torch's current stream) and the native collective
                sq = ShardedQuery(p, vec, native_all_reduce(comm, side.cuda_stream))
                assert sq.stream == side.cuda_stream
                r = sq.run()
                assert (r.n, r.converged, r.rounds, r.topup) == (w.n, w.converged, w.rounds, w.topup) and rel(r.sum, w.sum) <= 1e-13
                p.close()
            plans = [eng.plan(q) for q in qs[:3]]
            nb = Batch(plans)
            buf = torch.zeros(len(plans), max(p.totals_len for p in plans), dtype=torch.float64, device="cuda")
            for _ in range(3):
                comm.run_batch(nb, buf.data_ptr(), buf.shape[1], side.cuda_stream)
            for r, w in zip(nb.fetch(), want):
                assert (r.converged, r.rounds, r.topup_pending) == (w.converged, w.rounds, 1 if w.topup else 0)
                if not w.topup:
                    assert (r.n, r.visited) == (w.n, w.visited) and rel(r.sum, w.sum) <= 1e-13
            sb = ShardedBatch(plans, buf, native_all_reduce(comm, side.cuda_stream), batch=nb)  # default stream argument
            assert sb.stream == side.cuda_stream
            for r, w in zip(sb.run(), want):
                assert (r.n, r.converged, r.rounds, r.topup, r.topup_pending) == (w.n, w.converged, w.rounds, w.topup, 0) and rel(r.sum, w.sum) <= 1e-13
            nb.close()
            for p in plans:
                p.close()
        # on the legacy default stream the default argument is refused instead of racing with the collective
        p = eng.plan(qs[0])
        with pytest.raises(ValueError):
            ShardedQuery(p, torch.zeros(512, dtype=torch.float64, device="cuda"), lambda t_: None)
        p.close()
        comm.close()
        (c2,) = Comm.create_all([eng])  # the single-process form (ncclCommInitAll)
        assert (c2.nranks, c2.rank) == (1, 0)
        with torch.cuda.stream(side):
            t = torch.ones(8, dtype=torch.float64, device="cuda")
            c2.all_reduce_sum(t.data_ptr(), 8, side.cuda_stream)
            side.synchronize()
            assert float(t.sum()) == 8.0
        c2.close()


def test_plain_c_host_program(nat, tmp_path):
    """a plain-C host (gcc, no HIP headers, no Python in the data path) drives the C ABI — aqe_reduce, a
    batch of queries in ONE launch, and the sharded path through the library's own RCCL communicator."""
    import os, subprocess
    from approximatequeryengine_amd.build import LIB, ROOT
    exe = tmp_path / "host_demo"
    subprocess.check_call(["gcc", "-O1", "-Wall", "-Werror", "-std=c99", "-I", str(ROOT / "include"), str(ROOT / "tests" / "c_host" / "host_demo.c"),
                           "-o", str(exe), "-L", str(LIB.parent), "-laqe_hip", f"-Wl,-rpath,{LIB.parent}", "-lm"])
    env = dict(os.environ)  # (a process without torch: the system's HIP runtime and RCCL, one consistent pair)
    env["LD_LIBRARY_PATH"] = os.pathsep.join(["/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    out = subprocess.run([str(exe), "1000000"], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_demo ok" in out.stdout


def test_group_by_with_per_group_interval(nat, oracle, table):
    """GROUP BY region / product_id (executor.cpp:202-321): per-group (n, S, Q) against what SQLite returned for the
    reference's statements (tests/golden/groupby_sqlite.json), estimate and interval against the oracle's
    restatement; then any single-round sampler on a device-generated table against the oracle's grouping."""
    import json, os
    from approximatequeryengine_amd.engine import Engine, make_query
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "groupby_sqlite.json")))
    rows = table(g["table_rows"], g["seed"])
    col = {"region": nat.GROUP_REGION, "product_id": nat.GROUP_PRODUCT}
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=True)
        for case in g["cases"]:
            pct, where = case["sample_percent"], case["where"]
            want = [w for w in case["groups"] if w["count"] > 0]
            for agg, name in ((nat.SUM, "SUM"), (nat.AVG, "AVG"), (nat.COUNT, "COUNT")):
                got = eng.reduce_grouped(make_query(nat.M_ROWID_MOD, pct, agg=agg, where=tuple(where) if where else None), col[case["group_by"]])
                # (a key whose sampled rows all fail WHERE is listed with n = 0; SQLite's COUNT for it is 0 as well)
                assert [r.key for r in got if r.n] == [w["key"] for w in want] and all(r.visited for r in got)
                for r, w in zip([r for r in got if r.n], want):
                    assert r.n == w["count"] and rel(r.sum, w["sum"]) <= SUM_TOL and rel(r.sumsq, w["sumsq"]) <= SUM_TOL
                    v, lo, hi = oracle.group_ci(agg, w["count"], w["sum"], w["sumsq"], pct, reference_sum=False)
                    assert rel(r.value, v) <= EST_TOL and rel(r.ci_lower, lo) <= 1e-8 and rel(r.ci_upper, hi) <= 1e-8
                    if name == "AVG" and w["count"] >= 2:  # where the reference is right, the same numbers as the reference
                        rv, rlo, rhi = w["reference_ci"]["AVG"]
                        assert rel(r.value, rv) <= EST_TOL and rel(r.ci_lower, rlo) <= 1e-8 and rel(r.ci_upper, rhi) <= 1e-8
                    if name == "SUM" and w["count"] >= 2:  # the reference scales the MEAN: r.mean x scale is its "sum"
                        assert rel(r.mean * (100.0 / pct), w["reference_ci"]["SUM"][0]) <= EST_TOL
        with pytest.raises(nat.AqeError):  # multi-round (CLT) queries have no grouped form
            eng.reduce_grouped(make_query(nat.M_CLT_DUAL_POINTER, 20.0), nat.GROUP_REGION)
        eng.stage_records(rows, keep_aos=False)
        with pytest.raises(nat.AqeError):  # no key columns on the device
            eng.reduce_grouped(make_query(nat.M_ROWID_MOD, 10.0), nat.GROUP_REGION)
    n = 1_000_000
    big = table(n)
    with Engine(0) as eng:
        eng.generate_synthetic(n, keep_aos=False)  # keys follow from the row number
        for q, idx, where in (
            (make_query(nat.M_BLOCK, 5.0, where=(250.0, 750.0)), oracle.idx_block(n, 5.0, 1000), (250.0, 750.0)),
            (make_query(nat.M_MEMORY_STRIDE, 1.0), oracle.idx_memory_stride(n, 1.0), None),
            (make_query(nat.M_EXACT, 100.0), np.arange(n, dtype=np.uint64), None),
            (make_query(nat.M_PAGE, 3.0, block_size=4096), oracle.idx_page(n, 3.0, 4096), None),
        ):
            for column in (nat.GROUP_REGION, nat.GROUP_PRODUCT):
                got = eng.reduce_grouped(q, column)
                want = oracle.group(big, column, idx=idx, where=where)
                vis = dict((k, c) for k, c, *_ in oracle.group(big, column, idx=idx))
                want = {k: (c, s_, q_) for k, c, s_, q_ in want}
                assert [r.key for r in got] == sorted(vis)  # every key with a sampled row, WHERE or not
                for r in got:
                    c, s_, q_ = want.get(r.key, (0, 0.0, 0.0))
                    assert (r.n, r.visited) == (c, vis[r.key])
                    assert abs(r.sum - s_) <= SUM_TOL * max(abs(s_), 1.0) and abs(r.sumsq - q_) <= SUM_TOL * max(abs(q_), 1.0)
                assert sum(r.visited for r in got) == len(idx)


def test_group_by_key_ranges_nobody_picked_by_hand(nat, oracle):
    """GROUP BY over key columns of 1 ... 1024 distinct values, offset and negative minima, sparse keys — the register
    bins (<= 8 keys), the LDS bins, and their borders — with several samplers and WHERE, against the oracle's grouping."""
    from approximatequeryengine_amd.engine import Engine, make_query
    rng = np.random.default_rng(99)
    n = 200_003
    rows = oracle.synth(n, seed=5)
    for nb, kmin, sparse in ((1, 0, False), (2, -1, False), (3, 7, False), (5, -2, False), (7, 100, False), (8, 0, False), (9, 0, False),
                             (17, -8, False), (64, 1000, False), (1000, -500, False), (1024, 0, False), (40, 3, True)):
        keys = kmin + rng.integers(0, nb, n)
        if sparse:
            keys = kmin + (rng.integers(0, 4, n) * 13)  # four keys spread over a span of 40
        rows["region"] = keys.astype(np.int32)
        rows["product_id"] = (kmin + (np.arange(n) * 7) % nb).astype(np.int32)
        with Engine(0) as eng:
            eng.stage_records(rows, keep_aos=True)
            for q, idx, where in (
                (make_query(nat.M_EXACT, 100.0), np.arange(n, dtype=np.uint64), None),
                (make_query(nat.M_BLOCK, 7.0, block_size=333, where=(200.0, 900.0)), oracle.idx_block(n, 7.0, 333), (200.0, 900.0)),
                (make_query(nat.M_MEMORY_STRIDE, 3.0), oracle.idx_memory_stride(n, 3.0), None),
            ):
                for column in (nat.GROUP_REGION, nat.GROUP_PRODUCT):
                    got = eng.reduce_grouped(q, column, max_groups=1024)
                    want = {k: (c, s_, q_) for k, c, s_, q_ in oracle.group(rows, column, idx=idx, where=where, cap=4096)}
                    vis = dict((k, c) for k, c, *_ in oracle.group(rows, column, idx=idx, cap=4096))
                    assert [r.key for r in got] == sorted(vis), (nb, kmin, column)
                    for r in got:
                        c, s_, q_ = want.get(r.key, (0, 0.0, 0.0))
                        assert (r.n, r.visited) == (c, vis[r.key]), (nb, kmin, column, r.key)
                        # (sums are kept shifted by c ~ 500 and S, Q rebuilt from them: exact to rounding of n c and n c^2,
                        # which for a group of one small amount is coarser than rounding of S and Q themselves)
                        assert abs(r.sum - s_) <= SUM_TOL * max(abs(s_), 1e3 * max(c, 1)) and abs(r.sumsq - q_) <= SUM_TOL * max(abs(q_), 1e6 * max(c, 1))
                    assert sum(r.visited for r in got) == len(idx)


def test_group_by_over_virtual_shards(nat, oracle, table):
    """The multi-GPU form of GROUP BY on G shards of one GPU: agree the key range, bin per shard, add the bins
    (stand-in for the all-reduce), finish — identical to the single-shard answer for G in {1, 2, 3, 8}."""
    from approximatequeryengine_amd.distributed import sharded_group_by
    from approximatequeryengine_amd.engine import Engine, make_query
    import torch
    n = 100_007
    rows = table(n)
    queries = [make_query(nat.M_ROWID_MOD, 3.0, agg=nat.AVG), make_query(nat.M_BLOCK, 10.0, agg=nat.SUM, where=(250.0, 750.0)),
               make_query(nat.M_EXACT, 100.0, agg=nat.COUNT)]
    with Engine(0) as whole:
        whole.stage_records(rows, keep_aos=True)
        refs = {(i, col): whole.reduce_grouped(q, col) for i, q in enumerate(queries) for col in (nat.GROUP_REGION, nat.GROUP_PRODUCT)}
        # a world of one through the helper: identity collectives
        with torch.cuda.stream(torch.cuda.Stream()):  # the helper's default stream argument = torch's current stream
            bins = torch.zeros(4 * 1024, dtype=torch.float64, device="cuda")
            got = sharded_group_by(whole, queries[0], nat.GROUP_PRODUCT, bins, lambda t: None, lambda t: None)
        # (counts exact; a group's floating-point sum depends on the order lanes reach the LDS bin: equal to rounding)
        assert [(g_.key, g_.n) for g_ in got] == [(w.key, w.n) for w in refs[(0, nat.GROUP_PRODUCT)]]
        assert all(rel(g_.sum, w.sum) <= 1e-13 for g_, w in zip(got, refs[(0, nat.GROUP_PRODUCT)]))
    for G in (1, 2, 3, 8):
        bounds = [(g * n) // G for g in range(G + 1)]
        engs = []
        for g in range(G):
            e = Engine(0)
            e.stage_records(rows[bounds[g]:bounds[g + 1]], shard_lo=bounds[g], n_global=n, keep_aos=True)
            e.set_shift(float(rows["amount"][:1024].mean()))
            engs.append(e)
        side = torch.cuda.Stream()  # (a raw handle of 0 — torch's default stream — would mean "the context's own stream")
        st = side.cuda_stream
        for (i, col), want in refs.items():
            ranges = [e.group_key_range(col) for e in engs]
            kmin, kmax = min(r[0] for r in ranges), max(r[1] for r in ranges)
            nbins = kmax - kmin + 1
            with torch.cuda.stream(side):
                per = torch.full((G, 4 * nbins), float("nan"), dtype=torch.float64, device="cuda")
                for g, e in enumerate(engs):
                    e.grouped_enqueue_bins(queries[i], col, kmin, nbins, per[g].data_ptr(), st)
                total = per.sum(0)  # stand-in for the all-reduce, same stream => ordered
            for e in engs:  # every rank finishes from the same reduced bins
                got = e.grouped_finish(queries[i], kmin, nbins, total.data_ptr(), st)
                assert [g_.key for g_ in got] == [w.key for w in want]
                for g_, w in zip(got, want):
                    assert (g_.n, g_.visited) == (w.n, w.visited)
                    assert abs(g_.sum - w.sum) <= SUM_TOL * max(abs(w.sum), 1.0) and abs(g_.value - w.value) <= 1e-9 * max(abs(w.value), 1.0)
                    assert abs(g_.ci_lower - w.ci_lower) <= 1e-8 * max(abs(w.ci_lower), 1.0)
        for e in engs:
            e.close()


# ---- the variance-aware samplers over shards: the C-ABI pieces the ranks exchange (include/aqe_hip.h, distributed.py) ----
def test_plan_over_given_families(nat, oracle, engines, table):
    """aqe_plan_create_families: a query over the caller's families — a sampler's own families give the sampler's answer,
    runs of the amount-sorted column give the sum of those order statistics, AQE_M_EXACT leaves the sum unscaled, and a
    family that would leave the table never reaches a kernel."""
    from approximatequeryengine_amd.engine import make_query
    n = 200_003
    eng, rows = engines(n), table(n)
    q = make_query(nat.M_BLOCK, 5.0, block_size=700, where=(100.0, 900.0), agg=nat.AVG)
    fams, _, samples = nat.plan_families(q, n)
    want = eng.reduce(q)
    plan = eng.plan_families(q, fams, samples)
    plan.enqueue_all()
    got = plan.fetch()
    assert (got.n, got.visited) == (want.n, want.visited) and want.n > 0
    assert rel(got.sum, want.sum) <= SUM_TOL and rel(got.value, want.value) <= EST_TOL and rel(got.ci_upper, want.ci_upper) <= EST_TOL
    plan.close()
    # runs of the sorted column
    srt = np.sort(rows["amount"])
    runs = [(10, 500), (7_000, 1_234), (n - 3, 3), (50_000, 40_000)]
    fams = [nat.Family(row0=a, pitch=0, seg_len=l, step=1, ord_lo=0, ord_hi=l) for a, l in runs]
    take = np.concatenate([srt[a:a + l] for a, l in runs])
    for method, scaled in ((nat.M_EXACT, False), (nat.M_STRATIFIED_BLOCK, True)):
        qq = make_query(method, 100.0 * len(take) / n)
        plan = eng.plan_families(qq, fams, len(take), on_sorted=True)
        plan.enqueue_all()
        got = plan.fetch()
        assert got.n == len(take) and rel(got.sum, float(take.sum())) <= SUM_TOL
        if not scaled:
            assert rel(got.value, float(take.sum())) <= SUM_TOL
        plan.close()
    # refused on the host
    bad = [nat.Family(row0=n - 10, pitch=0, seg_len=11, step=1, ord_lo=0, ord_hi=11),
           nat.Family(row0=0, pitch=n, seg_len=4, step=1, ord_lo=0, ord_hi=8),
           nat.Family(row0=0, pitch=0, seg_len=1 << 40, step=1 << 30, ord_lo=0, ord_hi=1 << 35),
           nat.Family(row0=0, pitch=0, seg_len=8, step=1, ord_lo=0, ord_hi=8, flags=2),
           nat.Family(row0=0, pitch=0, seg_len=0, step=1, ord_lo=0, ord_hi=8),
           nat.Family(row0=n - 1200, pitch=1, seg_len=100, step=10, ord_lo=0, ord_hi=150)]  # overlapping segments: ordinal 99 is the largest row (and outside)
    for f in bad:
        for on_sorted in (False, True):
            with pytest.raises(nat.AqeError):
                eng.plan_families(q, [f], 8, on_sorted=on_sorted)


def test_zone_moments_and_sorted_counts_of_a_shard(nat, oracle, table):
    """aqe_zone_moments / aqe_sorted_counts on a shard in the middle of the table, against numpy on the same rows; a shard
    plans adaptive_block_sample only once the variances were agreed, and then exactly the whole-table plan's rows."""
    from approximatequeryengine_amd.engine import Engine, make_query
    from helpers import va_table
    n, lo, hi = 150_011, 40_000, 111_000
    rows = va_table(oracle, n, 0, n, ties=True)
    amt = rows["amount"]
    q = make_query(nat.M_ADAPTIVE_BLOCK, 5.0, block_size=200, block_size_max=900)
    with Engine(0) as eng:
        eng.stage_records(rows[lo:hi], shard_lo=lo, n_global=n)
        eng.set_shift(500.5)
        zm = eng.zone_moments()
        zs = n // 10
        for z in range(10):
            a, b = max(z * zs, lo), min((z + 1) * zs, hi)
            x = amt[a:b] if b > a else amt[:0]
            assert zm[z, 0] == len(x) and rel(zm[z, 1], float(x.sum())) <= SUM_TOL and rel(zm[z, 2], float((x * x).sum())) <= SUM_TOL
        srt = np.sort(amt[lo:hi])
        vals = np.concatenate([srt[::977], srt[::977] + 0.25, [-np.inf, np.inf, -1.0, 0.0, 1e9, srt[0], srt[-1]]])
        lt, le = eng.sorted_counts(vals)
        assert np.array_equal(lt, np.searchsorted(srt, vals, side="left")) and np.array_equal(le, np.searchsorted(srt, vals, side="right"))
        with pytest.raises(nat.AqeError):
            eng.plan(q)  # the shard does not know the other shards' zones
        var = np.zeros(10)
        for z in range(10):  # the reference's expression (DB.cpp:1291-1308)
            x = amt[z * zs:(z + 1) * zs]
            mean = x.sum() / len(x)
            var[z] = (x * x).sum() / len(x) - mean * mean
        eng.set_zone_variances(var)
        plan = eng.plan(q)
        plan.enqueue_all()
        got = plan.fetch()
        plan.close()
    idx = oracle.idx_adaptive_block(rows, 5.0, 200, 900)
    mine = idx[(idx >= lo) & (idx < hi)]
    assert len(mine) > 0 and got.n == len(mine) and rel(got.sum, float(amt[mine].sum())) <= SUM_TOL


def test_lognormal_table_against_the_reference(nat, oracle):
    """The skewed table of tests/golden/clt_lognormal.json (log-normal amounts: cv 2.9, single rows ~1 000 x the mean) on the
    device, against the reference's own recorded runs: the exact scan, the CLT monitor's stop in the race-free regime (T = 2:
    the leader stops on exactly the reference's row count), the T = 4 answers of the restatement, and five samplers — two of
    which (adaptive, stratified) pick their rows from the VALUES through the device pre-passes (zone variances, the sort)."""
    from approximatequeryengine_amd.engine import Engine, make_query
    from helpers import lognormal_table
    G, rows = lognormal_table(oracle)
    n = G["rows"]
    with Engine(0) as eng:
        eng.stage_records(rows, keep_aos=True)
        r = eng.reduce(make_query(nat.M_EXACT, 100.0))
        assert r.n == n and rel(r.sum, G["exact_sum"]) <= SUM_TOL
        for g in G["clt_fast_stop"]:
            for flags in (0, nat.Q_NO_PERSIST):
                q = make_query(nat.M_CLT_DUAL_POINTER, g["pct"], agg=nat.AVG, confidence_level=0.95, check_interval=g["check_interval"], num_threads=2,
                               max_error_percent=g["e"], flags=flags)
                r = eng.reduce(q)
                w = g["restatement"]
                assert r.converged == 1 and r.rounds == g["n_fast_at_stop"] // g["check_interval"], (g["e"], flags, r.rounds)
                assert (r.n, r.topup) == (w["n"], w["topup"]) and r.n - r.topup == 2 * g["n_fast_at_stop"]
                assert rel(r.value, w["avg"]) <= EST_TOL
        for c in G["clt_T4"]:
            w = c["restatement"]
            for flags in (0, nat.Q_NO_LEAN, nat.Q_NO_PERSIST):
                q = make_query(nat.M_CLT_DUAL_POINTER, c["pct"], agg=nat.AVG, confidence_level=0.95, check_interval=c["check_interval"], num_threads=c["T"],
                               max_error_percent=c["e"], flags=flags)
                r = eng.reduce(q)
                assert (r.n, r.topup, r.converged, r.rounds) == (w["n"], w["topup"], w["converged"], w["rounds"]), (c["e"], flags)
                assert rel(r.value, w["avg"]) <= EST_TOL
        for s in G["samplers"]:
            q = _query_for(nat, {"method": s["method"], "pct": s["pct"], "args": s["args"]})
            for flags in (0, nat.Q_FORCE_LEAN):
                q.flags = flags
                r = eng.reduce(q)
                assert r.n == s["n"] and rel(r.sum, s["sum"]) <= SUM_TOL and rel(r.sumsq, s["sumsq"]) <= SUM_TOL, (s["method"], flags)
            got = eng.gather(q)
            assert len(got) == s["n"]
            import hashlib
            if s["method"] != "stratified_block_sample":  # (gathered in the reference's order; the sorted sample's order among equal amounts is free)
                assert hashlib.sha256(np.ascontiguousarray(got["id"].astype(np.int64)).tobytes()).hexdigest() == s["ids_sha256"], s["method"]
            else:
                assert sorted(got["id"].tolist()) == sorted((oracle.idx_stratified_block(rows, s["pct"], int(s["args"][0]), int(s["args"][1])).astype(np.int64) + 1).tolist())


def test_peer_mapped_mailbox_all_reduce_in_one_process(nat, table):
    """aqe_mailbox_* with three contexts of ONE process (peers connected directly): every rank writes its vector into every
    mailbox and adds the slots up in rank order — the same sum on every rank, bit for bit, equal to the host's sum in that
    order, epoch after epoch; a whole sharded CLT query through it equals the single engine's; and a rank that never shows
    up ends the others' launches with its bit in the status instead of hanging the GPU."""
    import torch
    from approximatequeryengine_amd.distributed import ShardedQuery, mailbox_all_reduce, shard_bounds
    from approximatequeryengine_amd.engine import Engine, Mailbox, make_query
    G, n = 3, 300_007
    rows = table(n)
    engs = [Engine(0) for _ in range(G)]
    try:
        for g, e in enumerate(engs):
            lo, hi = shard_bounds(n, G, g)
            e.stage_records(rows[lo:hi], shard_lo=lo, n_global=n)
            e.set_shift(500.5)
        mbs = [Mailbox(e, G, g) for g, e in enumerate(engs)]
        Mailbox.connect_local(mbs)
        rng = np.random.default_rng(5)
        streams = [torch.cuda.Stream() for _ in range(G)]
        for epoch in range(40):
            count = int(rng.integers(1, nat.MAILBOX_MAX_DOUBLES + 1)) if epoch % 4 else (1, 8, 255, nat.MAILBOX_MAX_DOUBLES)[epoch // 4 % 4]
            host = [rng.standard_normal(count) * 10.0 ** rng.integers(-3, 6) for _ in range(G)]
            dev = [torch.from_numpy(h.copy()).cuda() for h in host]
            torch.cuda.synchronize()
            for g in range(G):
                mbs[g].all_reduce_sum(dev[g].data_ptr(), count, streams[g].cuda_stream)
            torch.cuda.synchronize()
            want = np.zeros(count)
            for h in host:
                want = want + h  # rank order
            for g in range(G):
                assert np.array_equal(dev[g].cpu().numpy(), want), (epoch, g)
            assert all(m.late_ranks() == 0 for m in mbs)
        with pytest.raises(nat.AqeError):
            mbs[0].all_reduce_sum(dev[0].data_ptr(), nat.MAILBOX_MAX_DOUBLES + 1, 0)
        # a sharded CLT query, stepwise, every collective through the mailboxes: the ranks run in threads (each blocks in its
        # own fetches), the launches meet on the device
        import threading
        q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, num_threads=4, max_error_percent=1.0, clt_round0=256, clt_growth=2)
        with Engine(0) as whole:
            whole.stage_records(rows)
            whole.set_shift(500.5)
            want = whole.reduce(q)
        got, errs = [None] * G, []

        def rank_main(g):
            try:
                torch.cuda.set_device(0)
                plan = engs[g].plan(q)
                st = streams[g].cuda_stream
                vec = torch.zeros(max(nat.MOMENT_VEC, plan.totals_len), dtype=torch.float64, device="cuda")
                torch.cuda.current_stream().synchronize()
                # (the stream handle is passed explicitly and torch is NOT switched to it: ShardedQuery orders what it does through
                # torch — zeroing the vector — on that stream itself)
                for batched in (False, True):
                    got[g] = ShardedQuery(plan, vec, mailbox_all_reduce(mbs[g], st), stream=st, batched=batched).run()
                    r = got[g]
                    assert (r.n, r.converged, r.rounds, r.topup) == (want.n, want.converged, want.rounds, want.topup)
                    assert rel(r.sum, want.sum) <= SUM_TOL and rel(r.value, want.value) <= EST_TOL
                plan.close()
            except Exception as ex:  # noqa: BLE001
                errs.append((g, repr(ex)))

        ts = [threading.Thread(target=rank_main, args=(g,)) for g in range(G)]
        for t in ts:
            t.start()
        for t in ts:
            t.join(timeout=120)
        assert not errs, errs
        assert all(m.late_ranks() == 0 for m in mbs)
        assert all((r.sum, r.value, r.ci_lower) == (got[0].sum, got[0].value, got[0].ci_lower) for r in got)  # bit for bit
        # rank 2 stays away: ranks 0 and 1 give up after the bound, their vectors untouched, bit 2 set
        keep = [torch.full((8,), float(g + 1), dtype=torch.float64, device="cuda") for g in range(2)]
        torch.cuda.synchronize()
        for g in range(2):
            mbs[g].all_reduce_sum(keep[g].data_ptr(), 8, streams[g].cuda_stream)
        torch.cuda.synchronize()
        for g in range(2):
            assert mbs[g].late_ranks() == 1 << 2 and np.array_equal(keep[g].cpu().numpy(), np.full(8, float(g + 1)))
    finally:
        for e in engs:
            e.close()


def test_key_range_counts_of_a_shard(nat, table):
    """aqe_key_range_counts: the rows of a shard below id_min / up to id_max — dense ids (arithmetic) and ids with gaps (two
    bisections over the resident rows) against numpy; the counts of the shards of a partition add up to the whole table's
    window (aqe_key_range_rows)."""
    from approximatequeryengine_amd.engine import Engine
    n = 90_001
    rows = table(n).copy()
    cuts = [0, 20_000, 20_001, 65_000, n]
    probes = [(-5, 10), (1, 1), (1, n), (20_000, 20_002), (30_000, 29_000), (n, n + 50), (n + 1, n + 9), (-2**63, 2**63 - 1), (64_999, 2**63 - 1)]
    for gaps in (False, True):
        if gaps:
            rows["id"] = 7 + 3 * np.arange(n) + (np.arange(n) % 2)  # ascending, not dense
        ids = rows["id"]
        pr = probes if not gaps else probes + [(7, 7), (8, 9), (int(ids[40_000]), int(ids[40_000])), (int(ids[-1]) + 1, 2**62)]
        with Engine(0) as whole:
            whole.stage_records(rows, keep_aos=True)
            want = [whole.key_range_rows(a, b) for a, b in pr]
        for (a, b), w in zip(pr, want):
            lo, hi = int(np.searchsorted(ids, a, side="left")), int(np.searchsorted(ids, b, side="right"))
            assert w == ((lo, hi) if hi > lo else (w[0], w[0])) or (hi <= lo and w[0] == w[1])
        total = [[0, 0] for _ in pr]
        for s0, s1 in zip(cuts[:-1], cuts[1:]):
            with Engine(0) as eng:
                eng.stage_records(rows[s0:s1], shard_lo=s0, n_global=n, keep_aos=True)
                for k, (a, b) in enumerate(pr):
                    below, upto = eng.key_range_counts(a, b)
                    sub = ids[s0:s1]
                    assert below == int(np.searchsorted(sub, a, side="left")), (gaps, s0, a, b)
                    assert upto == max(below, int(np.searchsorted(sub, b, side="right"))), (gaps, s0, a, b)
                    total[k][0] += below
                    total[k][1] += upto
        for (a, b), t, w in zip(pr, total, want):
            if w[1] > w[0]:
                assert tuple(t) == w, (gaps, a, b, t, w)
            else:
                assert t[1] <= t[0] or t[1] == t[0]
