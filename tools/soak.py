"""Dev helper: soak the persistent sweep — several query shapes interleaved on many streams, results checked against
the first execution of each plan.  A protocol failure shows up as AqeError (device_status) or a changed answer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
eng = Engine(0)
eng.generate_synthetic(10_000_000)
shapes = [dict(max_error_percent=0.01, clt_round0=4096, clt_growth=4), dict(max_error_percent=1.0, clt_round0=4096, clt_growth=4),
          dict(max_error_percent=0.05, clt_round0=512, clt_growth=2, num_threads=8), dict(max_error_percent=0.0, clt_round0=64, clt_growth=3, num_threads=6)]
plans, streams = [], []
for k in range(16):
    plans.append(eng.plan(make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, **shapes[k % len(shapes)])))
    streams.append(torch.cuda.Stream())
first = None
t0 = time.perf_counter()
for it in range(iters):
    for p, s in zip(plans, streams):
        p.enqueue_all(s.cuda_stream)
    if it % 200 == 199 or it == iters - 1:
        got = [(r.n, r.converged, r.rounds, r.topup, r.sum) for r in (p.fetch(s.cuda_stream) for p, s in zip(plans, streams))]
        if first is None:
            first = got
        assert got == first, (it, [a for a, b in zip(got, first) if a != b][:2])
dt = time.perf_counter() - t0
print(f"soak ok: {iters * len(plans)} queries in {dt:.1f} s ({iters * len(plans) / dt:.0f}/s), shapes: {[(g[0], g[1], g[2], g[3]) for g in first[:4]]}")
