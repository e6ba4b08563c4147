"""Dev helper (GPU box): single-round plans through the lean launch against k_round (AQE_Q_NO_LEAN), same box, same table.
    python tools/ab_single_round.py [rows ...]"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query

sizes = [int(x) for x in sys.argv[1:]] or [10_000_000, 100_000_000]
st = torch.cuda.Stream().cuda_stream
eng = Engine(0)
for rows in sizes:
    eng.generate_synthetic(rows)
    cases = [("exact", make_query(nat.M_EXACT, 100.0)), ("exact WHERE", make_query(nat.M_EXACT, 100.0, where=(250.0, 750.0))),
             ("stride 20%", make_query(nat.M_MEMORY_STRIDE, 20.0)), ("block 20%", make_query(nat.M_BLOCK, 20.0)),
             ("stride 1%", make_query(nat.M_MEMORY_STRIDE, 1.0)), ("block 1% WHERE", make_query(nat.M_BLOCK, 1.0, where=(250.0, 750.0), convention=nat.EST_CPP)),
             ("block 5% B=4096", make_query(nat.M_BLOCK, 5.0, block_size=4096)),
             ("CLT e=0.01%", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4))]
    for name, q in cases:
        line = []
        ref = None
        for flags in (0, nat.Q_NO_LEAN):
            q.flags = flags
            p = eng.plan(q)
            for _ in range(5):
                p.enqueue_all(st); r = p.fetch(st)
            p.set_profiling(True)
            us = []
            for _ in range(30):
                p.enqueue_all(st); r = p.fetch(st)
                us.append(1e3 * sum(p.launch_ms()))
            p.set_profiling(False)
            lat = []
            for _ in range(30):
                t0 = time.perf_counter(); p.enqueue_all(st); r = p.fetch(st); lat.append(1e6 * (time.perf_counter() - t0))
            k = nat.KERNEL_NAMES.get(p.last_kernel())
            if ref is None:
                ref = r
            else:
                assert (r.n, r.visited) == (ref.n, ref.visited) and abs(r.sum - ref.sum) <= 1e-11 * abs(ref.sum), (name, r.as_dict(), ref.as_dict())
            med = statistics.median(us)
            line.append("%s %.2f us (min %.2f) frac %.3f loop %.1f" % (k, med, min(us), 8.0 * r.visited / (med * 1e-6) / 8e12, statistics.median(lat)))
            p.close()
        print("%dM %-16s | %s" % (rows // 1_000_000, name, " | ".join(line)), flush=True)
eng.close()
