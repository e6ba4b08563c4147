"""oracle/make_golden_clt_hetero.py — TEST INFRASTRUCTURE ONLY.  The CLT monitor on a HETEROGENEOUS table, from the
reference's own C++ (oracle/_ref): the two halves of the table have different spreads, so the reference's fast threads —
each judging its own samples (custom_bplus_db.cpp:936-961), whichever gets there first raising should_stop — converge at
very different row counts.  The restatement names fast worker 0 the leader and lets only it judge: this fixture records
what the reference does where that choice matters (T = 4: two fast threads, T = 8: four), so that the tests can say
exactly how far the two are apart instead of claiming parity.

    python oracle/make_golden_clt_hetero.py        ->  tests/golden/clt_hetero.json
Table: the seeded synthetic table of N rows with amount' = 500.5 + (amount - 500.5) * s(row), s = 1 on the first half,
0.2 on the second (cv 0.576 / 0.115)."""
import json
import math
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.make_golden import SEED  # noqa: E402
from oracle.pyoracle import Oracle, Ref, build  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "clt_hetero.json"
N = 1_000_000
SCALES = (1.0, 0.2)


def hetero_rows(o, n=N, seed=SEED, flip=False):
    rows = o.synth(n, seed)
    s = np.where(np.arange(n) < n // 2, SCALES[1] if flip else SCALES[0], SCALES[0] if flip else SCALES[1])
    rows["amount"] = 500.5 + (rows["amount"] - 500.5) * s
    return rows


def main():
    build(ref=True)
    o = Oracle()
    G = {"rows": N, "seed": SEED, "scales": list(SCALES), "cases": []}
    for flip in (False, True):
        rows = hetero_rows(o, flip=flip)
        r = Ref()
        r.fill_direct(rows)
        for T in (4, 8):
            for e in (1.0, 0.5):
                runs = []
                for _ in range(30):
                    ids = r.sample("clt_validated_dual_pointer_sample", 20.0, 0.95, 10, T, e)
                    amt = r.last_amounts(len(ids))
                    runs.append({"n": int(len(ids)), "avg": math.fsum(amt) / len(ids)})
                rc, res, _ = o.clt_run(rows, 20.0, 0.95, 10, T, e)
                assert rc == 0
                case = {"flip": flip, "T": T, "e": e, "pct": 20.0, "check_interval": 10, "reference_runs": runs,
                        "true_mean": float(rows["amount"].mean()),
                        "restatement": {"n": int(res.final.n), "topup": int(res.topup), "converged": int(res.converged), "rounds": int(res.rounds),
                                        "leader_rows": int(res.fast.n), "avg": res.final.sum / res.final.n}}
                G["cases"].append(case)
                ns = [x["n"] for x in runs]
                print(f"flip={flip} T={T} e={e}: reference n {min(ns)}..{max(ns)} (median {sorted(ns)[15]}); restatement n {res.final.n} topup {res.topup} "
                      f"rounds {res.rounds} code {res.converged}", flush=True)
        r.close()
    OUT.write_text(json.dumps(G, indent=1))
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main()
