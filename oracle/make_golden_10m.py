"""oracle/make_golden_10m.py — TEST INFRASTRUCTURE ONLY.  Extends the 10 M-row table of tests/golden/ref_golden.json (the
BASELINE.json config size) from five calls to the whole deterministic suite, by running the reference's own C++
(oracle/_ref/libaqe_ref.so, compiled from /root/reference by oracle/Makefile) exactly as oracle/make_golden.py does
for the smaller tables; every other entry of the file is left byte for byte as it is.

    python oracle/make_golden_10m.py        (only where /root/reference was present at build time)

Per call: the index-set digest, exactly-rounded sums (math.fsum), the CLI's expressions on the returned amounts
(enhanced_aqe_cli.py:189-200, 277-291) for samples of at most 400 k rows, and a WHERE count/sum for block_sample.
The never-converging CLT call (max_error_percent = 0: 4 M rows, a deterministic multiset) takes the
reference about a minute here — its monitor re-scans every sample at every check (custom_bplus_db.cpp:936-946).
"""
import json
import math
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.make_golden import OUT, SEED, record_call  # noqa: E402
from oracle.pyoracle import Oracle, Ref, build  # noqa: E402


def main():
    build(ref=True)
    o = Oracle()
    N = 10_000_000
    rows = o.synth(N, SEED)
    r = Ref()
    r.fill_direct(rows)
    G = json.loads(OUT.read_text())
    big = {"N": N, "exact_sum": r.sum_amount(), "fsum": math.fsum(rows["amount"]), "calls": []}
    assert big["exact_sum"] == G["tables"][str(N)]["exact_sum"]
    suite = [
        ("memory_stride_sample", 1.0, (0,), False, True), ("random_pointer_sample", 1.0, (42,), False, True),
        ("block_sample", 1.0, (1000,), False, True), ("memory_stride_sample", 20.0, (0,), False, False),
        ("optimized_clt_sample", 20.0, (0.95, 20, 4, 2.0), False, False),
        # the rest of make_golden.deterministic_suite at the headline size
        ("memory_stride_sample", 1.0, (4096,), False, True), ("optimized_address_arithmetic_sample", 1.0, (), False, True),
        ("block_sample", 1.0, (77,), False, True), ("page_sample", 1.0, (4096,), False, True),
        ("parallel_block_sample", 1.0, (1000, 4), False, True), ("parallel_block_sample", 1.0, (300, 3), False, True),
        ("optimized_clt_sample", 1.0, (0.95, 20, 7, 2.0), False, True),
        ("fast_pointer_sample", 1.0, (2,), False, True), ("slow_pointer_sample", 1.0, (), False, True),
        ("dual_pointer_sample", 1.0, (), False, True), ("parallel_pointer_sample", 1.0, (4,), False, True),
        ("adaptive_block_sample", 1.0, (500, 2000), False, True), ("stratified_block_sample", 1.0, (1000, 4), False, True),
        ("random_pointer_sample", 1.0, (2147483647,), False, True),
        # the bench query's sampler, never converging: the exact multiset of 2 x base rows
        ("clt_validated_dual_pointer_sample", 20.0, (0.95, 10, 4, 0.0), True, False),
    ]
    for m, pct, args, srt, cli in suite:
        t0 = time.perf_counter()
        e = record_call(r, rows, m, pct, args, sort=srt, with_cli=cli, where=(250.0, 750.0) if (m == "block_sample" and args == (1000,)) else None)
        big["calls"].append(e)
        print(f"{m}{args} pct {pct}: n = {e.get('idx', {}).get('n')}  ({time.perf_counter() - t0:.1f} s)", flush=True)
    r.close()
    old = G["tables"][str(N)]["calls"]
    for a, b in zip(old, big["calls"]):  # the five calls already recorded must come out the same
        assert a == b, (a["method"], "changed")
    G["tables"][str(N)] = big
    OUT.write_text(json.dumps(G, indent=1))
    print(f"wrote {OUT}: {len(big['calls'])} calls at N = 10 M")


if __name__ == "__main__":
    main()
