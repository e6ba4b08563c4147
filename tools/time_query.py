"""Dev helper (GPU box): launch time (dispatch events) and closed loop of a few sparse 1 % samplers on 10 M / 100 M rows."""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
st = torch.cuda.Stream().cuda_stream
eng = Engine(0)
for rows in [int(x) for x in sys.argv[1:]] or [10_000_000]:
    eng.generate_synthetic(rows)
    for name, q in (("random device 1%", make_query(nat.M_RANDOM_DEVICE, 1.0, seed=42)), ("random host list 1%", make_query(nat.M_RANDOM_POINTER, 1.0, seed=42)),
                    ("stride 1%", make_query(nat.M_MEMORY_STRIDE, 1.0)), ("block 1%", make_query(nat.M_BLOCK, 1.0))):
        p = eng.plan(q)
        for _ in range(5):
            p.enqueue_all(st); r = p.fetch(st)
        p.set_profiling(True)
        us = []
        for _ in range(30):
            p.enqueue_all(st); r = p.fetch(st); us.append(1e3 * sum(p.launch_ms()))
        p.set_profiling(False)
        lat = []
        for _ in range(50):
            t0 = time.perf_counter(); p.enqueue_all(st); r = p.fetch(st); lat.append(1e6 * (time.perf_counter() - t0))
        print("%dM %-20s %-12s launch %.2f us (min %.2f) loop %.1f us  n=%d sum=%.6g" % (rows // 1_000_000, name, nat.KERNEL_NAMES.get(p.last_kernel()), statistics.median(us), min(us),
                                                                                 statistics.median(lat), r.n, r.sum), flush=True)
        p.close()
