"""Run under `rocprofv3 --pmc FETCH_SIZE`: k_sweep_multi where the queries of a batch cannot share a byte — 32 exact
scans over disjoint 10 M-row key ranges of a 320 M-row table (2.56 GB per launch, far beyond the Infinity Cache)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Batch, Engine, make_query
eng = Engine(0)
eng.generate_synthetic(320_000_000)
plans = [eng.plan(make_query(nat.M_EXACT, 100.0, agg=(nat.SUM, nat.AVG)[i % 2], rows=(10_000_000 * i, 10_000_000 * (i + 1)))) for i in range(32)]
b = Batch(plans)
for _ in range(12):
    b.enqueue_all(0)
    r = b.fetch()
print("done", r[0].value)
