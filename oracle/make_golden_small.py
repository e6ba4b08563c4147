"""oracle/make_golden_small.py — TEST INFRASTRUCTURE ONLY.  The two samplers the reference CLI routes SMALL tables to
(enhanced_aqe_cli.py:181-186), from the reference's own C++ (oracle/_ref) on tables populated through its public
insert_record path — these samplers see the real B+ tree, so the one-leaf fill of the other fixtures does not do:
  * leaf sizes after N ascending inserts (custom_bplus_db.cpp:164-240 + split, 43-62);
  * direct_access_sample(pct) (custom_bplus_db.cpp:584-644): deterministic — the exact rows;
  * optimized_sequential_sample(pct) (custom_bplus_db.cpp:366-428): its start offset comes from std::random_device — 12 runs
    per case, the rows of each (statistical parity: every run must be what the restatement gives for SOME start in [0, step));
  * std::uniform_real_distribution<double>(0, step) over std::mt19937(seed) of this container's libstdc++ — what the
    seeded variant of that start offset must equal.

    python oracle/make_golden_small.py      ->  tests/golden/small_tables.json"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tests"))
from helpers import digest  # noqa: E402
from oracle.make_golden import SEED  # noqa: E402
from oracle.pyoracle import Oracle, Ref, build  # noqa: E402

OUT = Path(__file__).resolve().parent.parent / "tests" / "golden" / "small_tables.json"
SIZES = (100, 254, 255, 300, 2_000, 9_999, 10_001, 20_000, 50_000)
PCTS = (0.5, 1.0, 5.0, 10.0, 20.0, 37.5, 100.0)


def main():
    build(ref=True)
    o = Oracle()
    G = {"seed": SEED, "tables": {}, "uniform_real": []}
    for seed in (0, 1, 42, 12345, 2**31 - 1, 2**32 - 1):
        for hi in (1.0, 10.0, 100.0 / 3.0, 200.0):
            r = Ref()
            G["uniform_real"].append({"seed": seed, "hi": hi, "value": r.uniform_real(seed, hi)})
            r.close()
    for n in SIZES:
        rows = o.synth(n, SEED)
        r = Ref()
        assert r.fill_insert(rows) == 0
        ls = r.leaf_sizes()
        T = {"leaves": int(len(ls)), "leaf_first": int(ls[0]), "leaf_last": int(ls[-1]), "leaf_sizes_distinct_inner": sorted(set(int(x) for x in ls[:-1])),
             "tree_height": r.tree_height(), "direct_access": [], "optimized_sequential": []}
        for pct in PCTS:
            ids = r.sample("direct_access_sample", pct)
            T["direct_access"].append({"pct": pct, "idx": digest(ids - 1), "distinct": int(len(np.unique(ids)))})
            runs = []
            for _ in range(12):
                ids = r.sample("optimized_sequential_sample", pct)
                runs.append([int(x) for x in (ids - 1)] if len(ids) <= 400 else {"n": int(len(ids)), "first": [int(x) for x in (ids - 1)[:64]], "last": int(ids[-1] - 1)})
            T["optimized_sequential"].append({"pct": pct, "runs": runs})
        r.close()
        G["tables"][str(n)] = T
        print(n, T["leaves"], T["leaf_last"], [d["idx"]["n"] for d in T["direct_access"]], flush=True)
    OUT.write_text(json.dumps(G))
    print(f"wrote {OUT} ({OUT.stat().st_size} bytes)")


if __name__ == "__main__":
    main()
