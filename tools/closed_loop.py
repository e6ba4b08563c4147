"""Dev helper (GPU box): closed-loop latency (aqe_plan_enqueue_all + aqe_plan_fetch) of a few query shapes, 10 M rows."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
eng = Engine(0)
eng.generate_synthetic(10_000_000)
shapes = (("stride 1%", make_query(nat.M_MEMORY_STRIDE, 1.0)), ("block 1% WHERE", make_query(nat.M_BLOCK, 1.0, where=(250.0, 750.0))),
          ("random 1%", make_query(nat.M_RANDOM_POINTER, 1.0, seed=42)), ("stride 20%", make_query(nat.M_MEMORY_STRIDE, 20.0)),
          ("exact", make_query(nat.M_EXACT, 100.0)),
          ("CLT e=0.01%", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4)),
          ("CLT e=1%", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=4096, clt_growth=4)))
for name, q in shapes:
    p = eng.plan(q)
    ref = eng.reduce(q)
    lat = []
    for _ in range(300):
        t0 = time.perf_counter()
        p.enqueue_all()
        r = p.fetch()
        lat.append(time.perf_counter() - t0)
        assert (r.n, r.visited, r.sum) == (ref.n, ref.visited, ref.sum) or abs(r.sum - ref.sum) <= 1e-12 * abs(ref.sum), (name, r.n, ref.n)
    lat.sort()
    print(f"{name}: p50 {1e6 * lat[150]:.1f} us, min {1e6 * lat[0]:.1f} us")
