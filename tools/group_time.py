import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
eng = Engine(0); eng.generate_synthetic(10_000_000)
for col in (nat.GROUP_REGION, nat.GROUP_PRODUCT):
    for q, name in ((make_query(nat.M_ROWID_MOD, 10.0, agg=nat.AVG), "rowid 10%"), (make_query(nat.M_EXACT, 100.0, agg=nat.SUM), "exact"), (make_query(nat.M_BLOCK, 1.0), "block 1%")):
        for _ in range(5): eng.reduce_grouped(q, col)
        t0 = time.perf_counter()
        for _ in range(50): r = eng.reduce_grouped(q, col)
        print(col, name, f"{(time.perf_counter()-t0)/50*1e6:.1f} us per call, groups {len(r)}")
