"""Dev helper (GPU box): the bench batch (32 CLT e=0.01% queries in one launch) with its 7 different pointer counts, and with
ONE pointer count for all 32 (every query then sweeps the very same rows: all but one copy are L2 hits)."""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Batch, Engine, make_query
eng = Engine(0); eng.generate_synthetic(10_000_000)
st = torch.cuda.Stream().cuda_stream
def run(name, qs):
    plans = [eng.plan(q) for q in qs]; b = Batch(plans)
    for _ in range(5): b.enqueue_all(st); b.fetch()
    b.set_profiling(True); ms = []
    for _ in range(50): b.enqueue_all(st); b.fetch(); ms.append(b.launch_info()[0])
    b.set_profiling(False)
    print("%-40s launch %.2f us (min %.2f)" % (name, 1e3 * statistics.median(ms), 1e3 * min(ms)), flush=True)
    b.close(); [p.close() for p in plans]
run("bench mix (T = 4, 6, ..., 16)", bench.headline_queries(nat, make_query, 32, 1, 0.01))
for T in (4, 8, 16):
    run(f"all 32 with T = {T}", [make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=(nat.AVG, nat.SUM, nat.COUNT)[i % 3], num_threads=T, max_error_percent=0.01 * (1 + 1e-3 * i),
                                          clt_round0=4096, clt_growth=4) for i in range(32)])
