"""Run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE in a second pass): a dense exact scan of the 10 M-row
amount column (known traffic: 80 MB of 8-byte-per-lane coalesced loads) to calibrate the counter for this
library's access width, then the single-query launch (k_sweep_lean), the exact scan as a lean launch, the bench step (ONE
k_sweep_lean_multi launch for a batch of 32 queries, in both layouts), a block sample, the seeded random sample (k_indexed: sector traffic of a sparse gather)
and the grouped reductions, so that each kernel's traffic can be read per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Batch, Engine, make_query
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
eng = Engine(0)
eng.generate_synthetic(n)
exact = make_query(nat.M_EXACT, 100.0)
exact.flags = nat.Q_NO_LEAN  # the calibration scan stays a k_round launch (its name + grid tell it apart in the counter file)
exact_lean = make_query(nat.M_EXACT, 100.0)
clt = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=bench.CLT_ROUND0, clt_growth=bench.CLT_GROWTH)
blk = make_query(nat.M_BLOCK, 1.0, block_size=1000)
rnd = make_query(nat.M_RANDOM_POINTER, 1.0, seed=42)
for _ in range(20):
    eng.reduce(exact)
for _ in range(40):
    eng.reduce(clt)
for _ in range(20):
    eng.reduce(exact_lean)  # the same 80 MB through k_sweep_lean: AFTER the 40 CLT launches of the same kernel name and grid (tools/pmc_summarize.py)
for _ in range(20):
    eng.reduce(blk)
for _ in range(20):
    eng.reduce(rnd)
for layout in ("", "packed"):  # the default (XCD-aligned) and the packed layout
    if layout:
        os.environ["AQE_MULTI_LAYOUT"] = layout
    plans = [eng.plan(q) for q in bench.headline_queries(nat, make_query, B, 1, 0.01)]
    b = Batch(plans)
    for _ in range(20):
        b.enqueue_all(0)
        b.fetch()
    b.close()
    for p in plans:
        p.close()
os.environ.pop("AQE_MULTI_LAYOUT", None)
for col in (nat.GROUP_PRODUCT, nat.GROUP_REGION):
    for q in (make_query(nat.M_ROWID_MOD, 10.0, agg=nat.AVG), make_query(nat.M_EXACT, 100.0, agg=nat.AVG)):
        for _ in range(20):
            eng.reduce_grouped(q, col)
print("done")
