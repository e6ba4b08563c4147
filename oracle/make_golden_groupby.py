"""Golden vectors for GROUP BY with per-group intervals (reference: src/aqe_backend/executor.cpp:202-321).

The reference's GROUP BY lives on its SQLite path, whose C++ cannot be built here (needs sqlite3 dev files).  What
that code does is build SQL text and hand it to SQLite, then do ten lines of arithmetic on (COUNT, SUM, SUM(x*x)).
This script issues the statements executor.cpp builds (lines 209-243) through SQLite itself — Python's stdlib
sqlite3 module, NOT a binary from the reference — on the synthetic `sales` table, and records SQLite's per-group
answers; the arithmetic of lines 247-302 is restated here in Python floats (executor_ci) and recorded beside
them, and the C restatement (oracle/aqe_oracle.c: aqo_group_ci) is checked against it.  tests/test_oracle_golden.py holds the oracle to this file.

    python oracle/make_golden_groupby.py      ->  tests/golden/groupby_sqlite.json
"""
import json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.pyoracle import Oracle

def statements(table, column, group_by, where, sample_percent):
    """The SQL of execute_query_groupby_with_ci: DISTINCT keys (EXE:209-212), then per key EXE:236-243."""
    step = 0 if (sample_percent <= 0 or sample_percent >= 100) else 100 // sample_percent  # EXE:21-26
    groups = f"SELECT DISTINCT {group_by} FROM {table}" + (f" WHERE {where}" if where else "")
    def per_group(gval):
        q = f"SELECT COUNT({column}), SUM({column}), SUM({column} * {column}) FROM {table} WHERE {group_by} = '{gval}'"
        if where: q += f" AND {where}"
        if step > 0: q += f" AND rowid % {step} = 0"
        return q
    return groups, per_group

def executor_ci(agg, count, total, sumsq, sample_percent):
    """executor.cpp:247-302 in Python floats (a second, independent restatement: the C one is checked against it)."""
    step = 0 if (sample_percent <= 0 or sample_percent >= 100) else 100 // sample_percent
    scale = 100.0 / sample_percent
    if count < 2:                                   # EXE:248-274
        v = total if agg == "SUM" else (total / count if count else 0.0)
        if step > 0 and agg != "AVG":
            v *= scale
        return [v, v, v]
    mean = total / count                            # EXE:277
    variance = (sumsq - (total * total / count)) / (count - 1)   # EXE:280
    margin = 1.96 * (variance / count) ** 0.5       # EXE:283-286
    if agg == "SUM":                                # EXE:289-296 (the MEAN is scaled, as the reference does)
        mean *= scale
        margin *= scale
    return [mean, mean - margin, mean + margin]


def main():
    o = Oracle()
    n = 20_000
    rows = o.synth(n, 42)
    db = sqlite3.connect(":memory:")
    db.execute("CREATE TABLE sales (id INTEGER PRIMARY KEY, amount REAL, region INTEGER, product_id INTEGER, timestamp INTEGER)")
    db.executemany("INSERT INTO sales VALUES (?,?,?,?,?)",
                   [(int(r["id"]), float(r["amount"]), int(r["region"]), int(r["product_id"]), int(r["timestamp"])) for r in rows])
    out = {"_about": __doc__.strip().split("\n\n")[0], "sqlite_version": sqlite3.sqlite_version, "table_rows": n, "seed": 42, "cases": []}
    for group_by in ("region", "product_id"):
        for pct in (10, 3, 100):
            for where in (None, "amount BETWEEN 250 AND 750"):
                gsql, per = statements("sales", "amount", group_by, where, pct)
                keys = sorted(int(r[0]) for r in db.execute(gsql))
                groups = []
                for k in keys:
                    c, s, q = db.execute(per(k)).fetchone()
                    c = int(c or 0)
                    s, q = float(s or 0.0), float(q or 0.0)
                    ci = {a: executor_ci(a, c, s, q, pct) for a in ("SUM", "AVG")}
                    groups.append({"key": k, "count": c, "sum": s, "sumsq": q, "reference_ci": ci})
                out["cases"].append({"group_by": group_by, "sample_percent": pct, "where": [250.0, 750.0] if where else None,
                                     "sql_groups": gsql, "sql_group_example": per(keys[0]), "groups": groups})
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "groupby_sqlite.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0)
    print(path, os.path.getsize(path), "bytes,", len(out["cases"]), "cases")

if __name__ == "__main__":
    main()
