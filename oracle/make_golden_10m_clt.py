"""oracle/make_golden_10m_clt.py — TEST INFRASTRUCTURE ONLY.  Adds to the 10 M-row table of tests/golden/ref_golden.json what
pins the CLT monitor in its CONVERGING regime at the bench's own size, from the reference's own C++ (oracle/_ref):
  * clt_fast_stop: where the reference's single fast worker stopped (T = 2 isolates it from the race, as
    oracle/make_golden.fast_stop_points does at 1 M rows) for e = 0.5 / 1 / 2 percent — the decision function
    (custom_bplus_db.cpp:936-961) on 10 M rows;
  * distributions.clt_e1_pct20_T4: 30 runs of the CLI's call at the other reading of the bench query
    (clt_validated_dual_pointer_sample(20, 0.95, 10, 4, 1.0)): rows returned and their mean — the racy regime, pinned as
    a range.
Every other entry of the file is left as it is.

    python oracle/make_golden_10m_clt.py
"""
import json
import math
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.make_golden import OUT, SEED, fast_stop_points  # noqa: E402
from oracle.pyoracle import Oracle, Ref, build  # noqa: E402


def main():
    build(ref=True)
    o = Oracle()
    N = 10_000_000
    rows = o.synth(N, SEED)
    r = Ref()
    r.fill_direct(rows)
    G = json.loads(OUT.read_text())
    T = G["tables"][str(N)]
    T["clt_fast_stop"] = fast_stop_points(r, rows, 20.0, 10, (0.5, 1.0, 2.0))
    print(T["clt_fast_stop"], flush=True)
    runs = []
    for i in range(30):
        ids = r.sample("clt_validated_dual_pointer_sample", 20.0, 0.95, 10, 4, 1.0)
        amt = r.last_amounts(len(ids))
        runs.append({"n": int(len(ids)), "avg": math.fsum(amt) / len(ids)})
        print(i, runs[-1], flush=True)
    T["distributions"] = {"clt_e1_pct20_T4": runs}
    r.close()
    OUT.write_text(json.dumps(G, indent=1))
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main()
