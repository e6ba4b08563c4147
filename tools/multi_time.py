"""Launch time of the one-launch batch (k_sweep_multi) by batch size Q: the bench query (10 M rows, CLT e = 0.01 %,
4 M samples = 32 MB per query), dispatch begin/end from the event pair on the launch, plus wall time per step with
every result fetched.  usage: python tools/multi_time.py [rows] [e]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Batch, Engine, make_query

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
e = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
eng = Engine(0)
eng.generate_synthetic(rows)
side = torch.cuda.Stream()
st = side.cuda_stream
for Q in (1, 2, 3, 4, 8, 16, 32, 64, 128):
    qs = [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=(nat.AVG, nat.SUM, nat.COUNT)[i % 3], num_threads=4 + 2 * (i % 7), max_error_percent=e,
                     clt_round0=4096, clt_growth=4) for i in range(Q)]
    plans = [eng.plan(q) for q in qs]
    b = Batch(plans)
    for _ in range(5):
        b.enqueue_all(st)
        r = b.fetch()
    b.set_profiling(True)
    ms = []
    for _ in range(30):
        b.enqueue_all(st)
        b.fetch()
        m, swept, wgs = b.launch_info()
        ms.append(m)
    b.set_profiling(False)
    ms.sort()
    k = ms[len(ms) // 2]
    t0 = time.perf_counter()
    steps = 100
    for _ in range(steps):
        b.enqueue_all(st)
        b.fetch()
    wall = (time.perf_counter() - t0) / steps
    # two batches in flight on two streams
    print(json.dumps({"Q": Q, "workgroups": wgs, "rows_swept": swept, "launch_us": round(1e3 * k, 2), "launch_us_min": round(1e3 * ms[0], 2),
                      "alg_GBps": round(8.0 * swept / (k * 1e-3) / 1e9, 1), "frac_of_8TBps": round(8.0 * swept / (k * 1e-3) / 8e12, 3),
                      "wall_us_per_step": round(1e6 * wall, 1), "aggregates_per_s": round(Q / wall), "n0": r[0].n, "rounds0": r[0].rounds,
                      "converged0": r[0].converged}), flush=True)
    b.close()
    for p in plans:
        p.close()
eng.close()
