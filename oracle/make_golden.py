#!/usr/bin/env python3
"""oracle/make_golden.py — TEST INFRASTRUCTURE ONLY.  Regenerates tests/golden/ref_golden.json.

Runs the reference's own C++ hot path (oracle/_ref/libaqe_ref.so, built by `make -C oracle ref` from
the sources under /root/reference) on the synthetic table of SURVEY §8d and records, per sampler call,
what the reference returned: sample count, first/last 16 row indices, CRC32 + wrapping sum of the whole
index list, and the exactly-rounded (math.fsum) sum and sum of squares of the sampled amounts.  It also
records the estimates/intervals obtained by pushing those samples through the reference CLI's
expressions (enhanced_aqe_cli.py:189-200, 262-291, restated below), decision points of the CLT monitor
observed on a single fast worker, run-to-run distributions of the nondeterministic entry points, the
façade's deterministic helpers, and a tiny DB file written by the reference's save_to_file.

Only this container can run it (the reference tree does not travel).  The JSON it writes is data:
inputs (N, seed, parameters) and the reference's outputs.

    python oracle/make_golden.py            # ~1 min
"""
from __future__ import annotations

import base64
import json
import math
import sys
import tempfile
import zlib
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.pyoracle import AVG, COUNT, SUM, Oracle, Ref, build  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "ref_golden.json"
SEED = 42


def digest(idx: np.ndarray) -> dict:
    idx = np.ascontiguousarray(idx, dtype="<u8")
    return {
        "n": int(len(idx)),
        "first": [int(x) for x in idx[:16]],
        "last": [int(x) for x in idx[-16:]],
        "crc32": zlib.crc32(idx.tobytes()) & 0xFFFFFFFF,
        "wsum": int(int(idx.sum(dtype=np.uint64)) % (1 << 64)),
    }


def cli_expressions(amounts: np.ndarray, total_records: int) -> dict:
    """enhanced_aqe_cli.py:189-200 (estimate) and :277-291 (interval), on a list of sampled amounts."""
    vals = [float(x) for x in amounts]
    n = len(vals)
    s = sum(vals)                                        # CLI:190 generator sum, list order
    out = {"py_sum": s, "SUM": s * (total_records / n),  # CLI:191-192
           "AVG": s / n, "COUNT": total_records}         # CLI:193-197
    if n > 1:
        mean = sum(vals) / len(vals)                     # CLI:279
        var = sum((x - mean) ** 2 for x in vals) / (len(vals) - 1)  # CLI:280
        std = var ** 0.5
        moe = 1.96 * std / (len(vals) ** 0.5)            # CLI:282
        scaled = moe * (total_records / n)               # CLI:286
        out.update({"moe": moe, "SUM_ci": [out["SUM"] - scaled, out["SUM"] + scaled],
                    "AVG_ci": [out["AVG"] - moe, out["AVG"] + moe]})
    return out


def record_call(ref: Ref, rows: np.ndarray, method: str, pct: float, args=(), sort=False,
                with_cli=True, where=None) -> dict:
    ids = ref.sample(method, pct, *args)
    entry = {"method": method, "pct": pct, "args": list(args)}
    if ids is None:
        entry["threw"] = True
        return entry
    idx = (ids - 1).astype(np.uint64)
    if sort:
        idx = np.sort(idx)
        entry["sorted"] = True
    entry["idx"] = digest(idx)
    amt = rows["amount"][idx.astype(np.int64)]
    entry["fsum"] = math.fsum(amt)
    entry["fsumsq"] = math.fsum(float(x) * float(x) for x in amt)
    if with_cli and len(idx) and len(idx) <= 400_000:
        entry["cli"] = cli_expressions(amt, len(rows))
    if where:
        keep = amt[(amt >= where[0]) & (amt <= where[1])]
        entry["where"] = {"range": list(where), "n": int(len(keep)), "fsum": math.fsum(keep)}
    return entry


def deterministic_suite(ref: Ref, rows: np.ndarray, pcts, with_clt=True) -> list:
    calls = []
    for pct in pcts:
        calls += [
            record_call(ref, rows, "memory_stride_sample", pct, (0,)),
            record_call(ref, rows, "memory_stride_sample", pct, (4096,)),
            record_call(ref, rows, "optimized_address_arithmetic_sample", pct),
            record_call(ref, rows, "random_pointer_sample", pct, (42,)),
            record_call(ref, rows, "block_sample", pct, (1000,), where=(250.0, 750.0)),
            record_call(ref, rows, "block_sample", pct, (77,)),
            record_call(ref, rows, "page_sample", pct, (4096,)),
            record_call(ref, rows, "parallel_block_sample", pct, (1000, 4)),
            record_call(ref, rows, "parallel_block_sample", pct, (300, 3)),
            record_call(ref, rows, "optimized_clt_sample", pct, (0.95, 20, 4, 2.0)),
            record_call(ref, rows, "optimized_clt_sample", pct, (0.95, 20, 7, 2.0)),
            record_call(ref, rows, "fast_pointer_sample", pct, (2,)),
            record_call(ref, rows, "slow_pointer_sample", pct),
            record_call(ref, rows, "dual_pointer_sample", pct),
            record_call(ref, rows, "parallel_pointer_sample", pct, (4,)),
            record_call(ref, rows, "adaptive_block_sample", pct, (500, 2000)),
            record_call(ref, rows, "adaptive_block_sample", pct, (64, 300), with_cli=False),
            record_call(ref, rows, "stratified_block_sample", pct, (1000, 4)),
            record_call(ref, rows, "stratified_block_sample", pct, (100, 5), with_cli=False),
        ]
        if with_clt:
            # max_error_percent = 0 never converges on non-constant data: deterministic multiset
            calls.append(record_call(ref, rows, "clt_validated_dual_pointer_sample", pct,
                                     (0.95, 10, 4, 0.0), sort=True))
            calls.append(record_call(ref, rows, "clt_validated_dual_pointer_sample", pct,
                                     (0.95, 10, 6, 0.0), sort=True, with_cli=False))
    return calls


def fast_stop_points(ref: Ref, rows: np.ndarray, pct: float, ci: int, es) -> list:
    """T=2 -> one fast worker over [0,N): its stop point is race-free as long as it stops before the
    slow rule's sample_count >= base/2 gate (DB.cpp:1011).  The fast chunk is the arithmetic progression
    from index 0 with the fast step (DB.cpp:925-927)."""
    N = len(rows)
    base = int(N * pct / 100.0)
    fast_step = max(3, int(N / (base // 1)))
    out = []
    for e in es:
        ids = ref.sample("clt_validated_dual_pointer_sample", pct, 0.95, ci, 2, e)
        idx = ids - 1
        # per-thread chunks are appended contiguously (DB.cpp:966-967); find the fast one
        starts = np.flatnonzero(idx == 0)
        n_fast = None
        for s in starts:
            k = 1
            while s + k < len(idx) and idx[s + k] == k * fast_step:
                k += 1
            if k > 1 or len(idx) == 1:
                n_fast = max(n_fast or 0, k)
        out.append({"pct": pct, "check_interval": ci, "T": 2, "e": e, "fast_step": fast_step,
                    "n_fast_at_stop": int(n_fast), "base": base, "returned": int(len(idx))})
    return out


def main():
    build(ref=True)
    o, ref_cls = Oracle(), Ref
    G = {"generator": {"name": "splitmix64 counter", "seed": SEED,
                       "amount": "1 + 999 * (splitmix64_at(seed, i) >> 11) * 2^-53",
                       "id": "i+1", "region": "i%4", "product_id": "i%100", "timestamp": "i"},
         "tables": {}}

    # raw std::mt19937 stream as exposed through random_pointer_sample is pinned by the index sets;
    # additionally pin the generator against numpy's legacy init_genrand seeding (same algorithm).
    G["mt19937_numpy_legacy"] = {
        str(s): [int(x) for x in np.random.RandomState(s)._bit_generator.random_raw(8)]
        for s in (0, 1, 42, 2**31 - 1)}

    for N, pcts in ((10_000, (1.0, 0.7, 20.0, 33.3, 100.0)), (100_000, (1.0, 5.0, 20.0)),
                    (1_000_000, (1.0, 20.0)), (100_007, (1.0, 20.0))):
        rows = o.synth(N, SEED)
        r = ref_cls()
        r.fill_direct(rows)
        T = {"N": N, "cache_rows": r.cache_rows(), "amount0": float(rows["amount"][0]),
             "exact": {"sum_amount": r.sum_amount(), "fsum": math.fsum(rows["amount"]),
                       "where": [{"range": [a, b], "sum": r.sum_amount_where(a, b),
                                  "fsum": math.fsum(rows["amount"][(rows["amount"] >= a) & (rows["amount"] <= b)])}
                                 for a, b in ((250.0, 750.0), (0.0, 10.0), (999.0, 99999.99))]},
             "calls": deterministic_suite(r, rows, pcts)}
        if N in (10_000, 1_000_000):
            T["random_seeds"] = [record_call(r, rows, "random_pointer_sample", 1.0, (s,), with_cli=False)
                                 for s in (0, 1, 42, 2**31 - 1)]
        if N == 1_000_000:
            T["clt_fast_stop"] = fast_stop_points(r, rows, 20.0, 10, (0.5, 1.0, 2.0, 5.0)) + \
                fast_stop_points(r, rows, 10.0, 20, (1.0, 3.0))
            # run-to-run distributions of the racy / random_device-seeded entry points
            dist = {"clt_e1_pct20_T4": [], "fast_aggregated_pct1_T4": [], "parallel_sum_pct1_T4": [],
                    "parallel_sum_where_250_750_pct1_T4": [], "parallel_count_pct1_T4": []}
            for _ in range(30):
                ids = r.sample("clt_validated_dual_pointer_sample", 20.0, 0.95, 10, 4, 1.0)
                amt = rows["amount"][ids - 1]
                dist["clt_e1_pct20_T4"].append({"n": int(len(ids)), "avg": math.fsum(amt) / len(ids)})
                dist["fast_aggregated_pct1_T4"].append(r.fast_aggregated_memory_stride_sum(1.0, 4))
            for _ in range(10):
                dist["parallel_sum_pct1_T4"].append(r.parallel_sum_sample(1.0, 4))
                dist["parallel_sum_where_250_750_pct1_T4"].append(r.parallel_sum_where_sample(250, 750, 1.0, 4))
                dist["parallel_count_pct1_T4"].append(r.parallel_count_sample(1.0, 4))
            T["distributions"] = dist
            # random_start_memory_stride_sample (DB.cpp:1838-1878): the draw is visible as the first row
            rs = []
            for pct_, sb in ((1.0, 0), (5.0, 0), (1.0, 4096)):
                for _ in range(3):
                    ids = r.sample("random_start_memory_stride_sample", pct_, sb)
                    idx = (ids - 1).astype(np.uint64)
                    rs.append({"pct": pct_, "stride_bytes": sb, "start": int(idx[0]), "idx": digest(idx)})
            T["random_start_stride"] = rs
        r.close()
        G["tables"][str(N)] = T
        print(f"N={N}: {len(T['calls'])} calls", flush=True)

    # insert_record-built tree (faithful, quadratic) vs the O(N) direct fill: same outputs
    N = 3000
    rows = o.synth(N, SEED)
    a, b = ref_cls(), ref_cls()
    a.fill_insert(rows)
    b.fill_direct(rows)
    same = True
    for m, args in (("memory_stride_sample", (0,)), ("random_pointer_sample", (42,)), ("block_sample", (100,)),
                    ("optimized_clt_sample", (0.95, 20, 4, 2.0)), ("parallel_block_sample", (100, 4))):
        same &= bool((a.sample(m, 10.0, *args) == b.sample(m, 10.0, *args)).all())
    same &= a.sum_amount() == b.sum_amount() and a.cache_rows() == b.cache_rows()
    G["insert_vs_direct_3000"] = {"identical": same, "tree_height": a.tree_height(),
                                  "node_count": a.node_count(), "cache_rows": a.cache_rows()}
    # small-N quirks of the real tree: leaf root (N<255) vs internal root with an empty cache
    quirks = []
    for n_small in (100, 254, 300, 999, 1000, 1001):
        rs = o.synth(n_small, SEED)
        q = ref_cls()
        q.fill_insert(rs)
        ids = q.sample("memory_stride_sample", 10.0, 0)
        quirks.append({"N": n_small, "cache_rows_before": 0 if n_small < 1000 else 1000,
                       "memory_stride_n": int(len(ids)), "first": [int(x - 1) for x in ids[:4]],
                       "cache_rows_after": q.cache_rows(),
                       "block_n": int(len(q.sample("block_sample", 10.0, 10)))})
        q.close()
    G["small_n_quirks"] = quirks

    # file written by the reference's save_to_file (DB.cpp:665-683)
    few = o.synth(5, SEED)
    f = ref_cls()
    f.fill_insert(few)
    with tempfile.TemporaryDirectory() as d:
        p = Path(d) / "five.db"
        assert f.save_to_file(p)
        G["file_5_rows_b64"] = base64.b64encode(p.read_bytes()).decode()
    f.close()
    a.close()
    b.close()

    # façade helpers (SCH.cpp:277-305)
    any_ref = ref_cls()
    G["confidence"] = [{"pct": p, "N": n, "value": any_ref.sched_confidence(p, n)}
                       for p, n in ((10.0, 100_000), (1.0, 60_000), (1.0, 10_000), (0.5, 10_000), (1.0, 4_000),
                                    (0.1, 10_000), (10.0, 10_000_000))]
    G["where_parse"] = [{"query": q, "range": list(any_ref.sched_where(q))} for q in (
        "SELECT SUM(amount) FROM sales WHERE amount BETWEEN 250 AND 750",
        "SELECT SUM(amount) FROM sales WHERE amount >= 10.5 AND amount <= 99.25",
        "SELECT SUM(amount) FROM sales WHERE amount > 500",
        "SELECT SUM(amount) FROM sales WHERE amount>=5 AND amount<=6",
        "SELECT SUM(amount) FROM sales",
        "SELECT SUM(amount) FROM sales WHERE region = 2")]
    any_ref.close()

    # scalar-only results at the BASELINE config size (10 M rows)
    N = 10_000_000
    rows = o.synth(N, SEED)
    r = ref_cls()
    r.fill_direct(rows)
    big = {"N": N, "exact_sum": r.sum_amount(), "fsum": math.fsum(rows["amount"]), "calls": []}
    for m, pct, args, srt in (("memory_stride_sample", 1.0, (0,), False), ("random_pointer_sample", 1.0, (42,), False),
                              ("block_sample", 1.0, (1000,), False), ("memory_stride_sample", 20.0, (0,), False),
                              ("optimized_clt_sample", 20.0, (0.95, 20, 4, 2.0), False)):
        big["calls"].append(record_call(r, rows, m, pct, args, sort=srt, with_cli=(pct <= 1.0),
                                        where=(250.0, 750.0) if m == "block_sample" else None))
    r.close()
    G["tables"][str(N)] = big
    print("N=10M done", flush=True)

    OUT.parent.mkdir(parents=True, exist_ok=True)
    OUT.write_text(json.dumps(G, indent=1))
    print(f"wrote {OUT} ({OUT.stat().st_size/1024:.0f} KiB)")


if __name__ == "__main__":
    main()
