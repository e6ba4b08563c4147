"""Dev helper: the one-launch-per-round form (AQE_Q_NO_PERSIST, or plans of more than 32 rounds) replayed as a HIP graph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
eng = Engine(0)
for n, kw in ((1_000_000, dict(clt_round0=16, clt_growth=2, num_threads=8, max_error_percent=0.0)),
              (200_000, dict(max_error_percent=0.0)),            # the reference's cadence: check_interval = 10 per round
              (10_000_000, dict(clt_round0=4096, clt_growth=4, max_error_percent=0.01)),
              (10_000_000, dict(max_error_percent=2.0))):        # the mirror API's defaults on a big table: 100 000 rounds planned
    eng.generate_synthetic(n)
    q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, **kw)
    q.flags = nat.Q_NO_PERSIST
    r = eng.reduce(q)
    for _ in range(3): eng.reduce(q)
    t0 = time.perf_counter()
    for _ in range(20): r = eng.reduce(q)
    print(n, kw, "rounds", r.rounds, "n", r.n, f"{(time.perf_counter()-t0)/20*1e6:.1f} us per query")
