"""Times every BASELINE.json configuration that fits one GPU (kernel time from HIP events around the query's
launches, table resident in HBM) and prints one JSON line per case: algorithmic bytes = 8 B x samples."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query

def run(eng, name, q, reps=30):
    for _ in range(5):
        r = eng.reduce(q)
    ms = []
    t0 = time.perf_counter()
    for _ in range(reps):
        r = eng.reduce(q)
        ms.append(r.kernel_ms)
    wall = (time.perf_counter() - t0) / reps
    ms.sort()
    k = ms[len(ms) // 2]
    print(json.dumps({"case": name, "samples": r.visited, "n": r.n, "value": r.value, "ci": [r.ci_lower, r.ci_upper],
                      "converged": r.converged, "rounds": r.rounds, "kernel_us": round(1e3 * k, 2), "wall_us": round(1e6 * wall, 1),
                      "alg_GBps": round(8.0 * r.visited / (k * 1e-3) / 1e9, 1) if k > 0 else None}), flush=True)

sizes = [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["10000000", "100000000", "1000000000"])]
eng = Engine(0)
for n in sizes:
    t0 = time.perf_counter()
    eng.generate_synthetic(n)
    print(json.dumps({"table_rows": n, "generate_s": round(time.perf_counter() - t0, 3)}), flush=True)
    tag = f"{n // 1_000_000}M"
    run(eng, f"{tag} exact SUM (full scan)", make_query(nat.M_EXACT, 100.0))
    run(eng, f"{tag} exact SUM WHERE 250..750", make_query(nat.M_EXACT, 100.0, where=(250.0, 750.0)))
    run(eng, f"{tag} stride 1% SUM", make_query(nat.M_MEMORY_STRIDE, 1.0))
    run(eng, f"{tag} stride 20% SUM", make_query(nat.M_MEMORY_STRIDE, 20.0))
    if n <= 100_000_000:
        run(eng, f"{tag} random 1% seed 42 SUM", make_query(nat.M_RANDOM_POINTER, 1.0, seed=42), reps=10)
    run(eng, f"{tag} block 1% B=1000 SUM", make_query(nat.M_BLOCK, 1.0))
    run(eng, f"{tag} block 1% B=1000 SUM WHERE 250..750 (config 5)", make_query(nat.M_BLOCK, 1.0, where=(250.0, 750.0), convention=nat.EST_CPP))
    run(eng, f"{tag} block 20% B=1000 SUM", make_query(nat.M_BLOCK, 20.0))
    run(eng, f"{tag} page 5% SUM", make_query(nat.M_PAGE, 5.0, block_size=4096))
    for e in (0.01, 0.005, 1.0, 0.5):
        run(eng, f"{tag} CLT AVG e={e}% (R0=4096,g=4)", make_query(nat.M_CLT_DUAL_POINTER, nat.lib().aqe_error_to_sample_percent(e), agg=nat.AVG,
                                                             max_error_percent=e, clt_round0=4096, clt_growth=4))
eng.close()
