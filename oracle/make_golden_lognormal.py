"""oracle/make_golden_lognormal.py — TEST INFRASTRUCTURE ONLY.  The CLT monitor on a SKEWED table, from the reference's own
C++ (oracle/_ref): the second data set SURVEY 8d recommends for CLT behaviour — amount ~ log-normal(mu = 5, sigma = 1.5),
coefficient of variation ~2.9, a tail where a handful of rows carry a visible share of the sum — on which the running
mean and variance of a pointer move in jumps and the error rule (custom_bplus_db.cpp:936-961) is met, lost and met again.

    python oracle/make_golden_lognormal.py        ->  tests/golden/clt_lognormal.json
Recorded: where the reference's single fast thread stops (T = 2: race-free, exact), the row counts and averages of 30 runs
at T = 4, the strided and block samplers' sums on the same table (the estimators meet large values), and the restatement's
own answers.  The table is regenerated in the tests from numpy's PCG64 stream (seed below); its digest is in the fixture, so
a platform whose libm rounds exp() differently is detected rather than mis-compared."""
import hashlib
import json
import math
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle.make_golden import SEED, fast_stop_points  # noqa: E402
from oracle.pyoracle import Oracle, Ref, build  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "clt_lognormal.json"
N = 1_000_000
MU, SIGMA, RNG_SEED = 5.0, 1.5, 20260
SAMPLERS = [("memory_stride_sample", 1.0, (0,)), ("block_sample", 5.0, (1000,)), ("optimized_clt_sample", 20.0, (0.95, 20, 4, 2.0)),
            ("adaptive_block_sample", 5.0, (500, 2000)), ("stratified_block_sample", 5.0, (1000, 4))]  # (the last two follow the VALUES)


def lognormal_rows(o, n=N):
    rows = o.synth(n, SEED)
    rows["amount"] = np.random.Generator(np.random.PCG64(RNG_SEED)).lognormal(MU, SIGMA, n)
    return rows


def digest(rows):
    return hashlib.sha256(np.ascontiguousarray(rows["amount"]).tobytes()).hexdigest()


def main():
    build(ref=True)
    o = Oracle()
    rows = lognormal_rows(o)
    amt = rows["amount"]
    G = {"rows": N, "seed": SEED, "mu": MU, "sigma": SIGMA, "rng": "numpy PCG64", "rng_seed": RNG_SEED, "amount_sha256": digest(rows),
         "true_mean": float(amt.mean()), "cv": float(amt.std() / amt.mean()), "max_over_mean": float(amt.max() / amt.mean())}
    r = Ref()
    r.fill_direct(rows)
    G["exact_sum"] = r.sum_amount()
    G["clt_fast_stop"] = fast_stop_points(r, rows, 20.0, 10, (10.0, 5.0, 3.0))
    for g in G["clt_fast_stop"]:
        rc, res, _ = o.clt_run(rows, g["pct"], 0.95, g["check_interval"], 2, g["e"])
        assert rc == 0
        g["restatement"] = {"n": int(res.final.n), "topup": int(res.topup), "converged": int(res.converged), "rounds": int(res.rounds),
                            "leader_rows": int(res.fast.n), "avg": res.final.sum / res.final.n}
        print(f"T=2 e={g['e']}: reference fast thread stopped at {g['n_fast_at_stop']} rows; restatement leader {res.fast.n} rows, code {res.converged}", flush=True)
    G["clt_T4"] = []
    for e in (10.0, 5.0):
        runs = []
        for _ in range(30):
            ids = r.sample("clt_validated_dual_pointer_sample", 20.0, 0.95, 10, 4, e)
            a = r.last_amounts(len(ids))
            runs.append({"n": int(len(ids)), "avg": math.fsum(a) / len(ids)})
        rc, res, _ = o.clt_run(rows, 20.0, 0.95, 10, 4, e)
        assert rc == 0
        G["clt_T4"].append({"e": e, "pct": 20.0, "check_interval": 10, "T": 4, "reference_runs": runs,
                            "restatement": {"n": int(res.final.n), "topup": int(res.topup), "converged": int(res.converged), "rounds": int(res.rounds),
                                            "leader_rows": int(res.fast.n), "avg": res.final.sum / res.final.n}})
        ns = [x["n"] for x in runs]
        print(f"T=4 e={e}: reference n {min(ns)}..{max(ns)}; restatement n {res.final.n} topup {res.topup} rounds {res.rounds} code {res.converged}", flush=True)
    G["samplers"] = []
    for name, pct, args in SAMPLERS:
        ids = r.sample(name, pct, *[float(a) for a in args])
        a = r.last_amounts(len(ids))
        G["samplers"].append({"method": name, "pct": pct, "args": list(args), "n": int(len(ids)), "sum": math.fsum(a), "sumsq": math.fsum(x * x for x in a),
                              "ids_sha256": hashlib.sha256(np.ascontiguousarray(ids, dtype=np.int64).tobytes()).hexdigest()})
        print(f"{name} {pct}%: n {len(ids)} sum {math.fsum(a):.6g}", flush=True)
    r.close()
    OUT.write_text(json.dumps(G, indent=1))
    print(f"wrote {OUT}")


if __name__ == "__main__":
    main()
