"""Dev helper (GPU box): launch time (dispatch begin -> end, profiling events) of a handful of single queries under the
library AQE_HIP_LIB names: the bench CLT query (k_sweep_persist), exact scans and strided samples (k_round) on 10 M and
100 M rows.  One line per run, for A/B comparisons of kernel variants on one box."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
sizes = [int(x) for x in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["10000000", "100000000"])]
side = torch.cuda.Stream()
st = side.cuda_stream
out = [os.path.basename(os.environ.get("AQE_HIP_LIB", "default")) + " cap=" + os.environ.get("AQE_ROUND_MAX_BLOCKS", "-")]
for rows in sizes:
    eng = Engine(0)
    eng.generate_synthetic(rows)
    qs = [("clt", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4)),
          ("exact", make_query(nat.M_EXACT, 100.0)),
          ("exactW", make_query(nat.M_EXACT, 100.0, where=(250.0, 750.0))),
          ("s20", make_query(nat.M_MEMORY_STRIDE, 20.0)),
          ("s1", make_query(nat.M_MEMORY_STRIDE, 1.0)),
          ("b20", make_query(nat.M_BLOCK, 20.0, block_size=1000))]
    for name, q in qs:
        p = eng.plan(q)
        for _ in range(10):
            p.enqueue_all(st); p.fetch(st)
        p.set_profiling(True)
        ms = []
        for _ in range(n):
            p.enqueue_all(st)
            torch.cuda.synchronize()
            ms.append(sum(p.launch_ms()))
        p.set_profiling(False)
        r = p.fetch(st)
        out.append("%dM %s %.2f/%.2f" % (rows // 1000000, name, 1e3 * statistics.median(ms), 1e3 * min(ms)))
        p.close()
    eng.close()
print(" | ".join(out))
