import os, sys, time
sys.path.insert(0, os.getcwd())
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
eng = Engine(0); eng.generate_synthetic(10_000_000)
shapes = (("stride 1%", make_query(nat.M_MEMORY_STRIDE, 1.0)),
          ("CLT e=0.01%", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4)),
          ("CLT e=1%", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=4096, clt_growth=4)),
          ("reduce stride", None))
for name, q in shapes:
    if q is None:
        q = make_query(nat.M_BLOCK, 1.0)
        ref = eng.reduce(q); lat = []
        for _ in range(50000):
            t0 = time.perf_counter(); r = eng.reduce(q); lat.append(time.perf_counter() - t0)
            assert (r.n, r.sum) == (ref.n, ref.sum)
    else:
        p = eng.plan(q); ref = eng.reduce(q); lat = []
        for _ in range(50000):
            t0 = time.perf_counter(); p.enqueue_all(); r = p.fetch(); lat.append(time.perf_counter() - t0)
            assert (r.n, r.visited, r.rounds, r.converged) == (ref.n, ref.visited, ref.rounds, ref.converged) and abs(r.sum - ref.sum) <= 1e-12 * abs(ref.sum)
    lat.sort()
    print(f"{name}: p50 {1e6*lat[len(lat)//2]:.1f} p99 {1e6*lat[int(len(lat)*0.99)]:.1f} p99.99 {1e6*lat[int(len(lat)*0.9999)]:.1f} max {1e6*lat[-1]:.1f} us")
