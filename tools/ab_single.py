"""Dev helper (GPU box): launch time of the single-query persistent sweep (k_sweep_persist, bench query) under the
library AQE_HIP_LIB names: dispatch begin/end from the profiling events, mean / median / min over N launches."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
e = float(sys.argv[2]) if len(sys.argv) > 2 else 0.01
eng = Engine(0)
eng.generate_synthetic(10_000_000)
side = torch.cuda.Stream()
st = side.cuda_stream
q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=e, clt_round0=4096, clt_growth=4)
p = eng.plan(q)
for _ in range(20):
    p.enqueue_all(st); p.fetch(st)
p.set_profiling(True)
ms = []
for _ in range(n):
    p.enqueue_all(st)
    torch.cuda.synchronize()
    ms.append(sum(p.launch_ms()))
p.set_profiling(False)
import time
lat = []
for _ in range(0 if os.environ.get("AB_NO_LOOP") else 200):
    t0 = time.perf_counter(); p.enqueue_all(st); p.fetch(st); lat.append(time.perf_counter() - t0)
print(os.path.basename(os.environ.get("AQE_HIP_LIB", "default")), "launch us mean %.2f median %.2f min %.2f | closed loop p50 %.2f" % (
    1e3 * statistics.mean(ms), 1e3 * statistics.median(ms), 1e3 * min(ms), 1e6 * statistics.median(lat) if lat else -1.0))
