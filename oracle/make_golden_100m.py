"""oracle/make_golden_100m.py — TEST INFRASTRUCTURE ONLY.  Adds a 100 M-row table (BASELINE.json configs 2 and 4: "100M-row
APPROX SUM, 1% sampling" and "100M-row block-sampling method with WHERE range predicate") to
tests/golden/ref_golden.json by running the reference's own C++ (oracle/_ref/libaqe_ref.so) on the seeded synthetic
table, as oracle/make_golden.py does for the smaller ones; every other entry of the file is left as it is.
Needs about 16 GB of memory (the reference keeps the rows three times and copies them once more per call).

    python oracle/make_golden_100m.py
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
    N = 100_000_000
    rows = o.synth(N, SEED)
    r = Ref()
    r.fill_direct(rows)
    big = {"N": N, "exact_sum": r.sum_amount(), "calls": []}
    for m, pct, args, where in (("memory_stride_sample", 1.0, (0,), None), ("block_sample", 1.0, (1000,), (250.0, 750.0)),
                                ("random_pointer_sample", 1.0, (42,), None), ("page_sample", 1.0, (4096,), None),
                                ("parallel_block_sample", 1.0, (1000, 4), None), ("optimized_clt_sample", 1.0, (0.95, 20, 4, 2.0), None)):
        t0 = time.perf_counter()
        e = record_call(r, rows, m, pct, args, with_cli=False, where=where)
        big["calls"].append(e)
        print(f"{m}{args}: n = {e['idx']['n']}  ({time.perf_counter() - t0:.1f} s)", flush=True)
    r.close()
    G = json.loads(OUT.read_text())
    G["tables"][str(N)] = big
    OUT.write_text(json.dumps(G, indent=1))
    print(f"wrote {OUT}: {len(big['calls'])} calls at N = 100 M")


if __name__ == "__main__":
    main()
