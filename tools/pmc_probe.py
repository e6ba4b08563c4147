"""Run under `rocprofv3 --pmc FETCH_SIZE` (and WRITE_SIZE in a second pass): a dense exact scan of the 10 M-row
amount column (known traffic: 80 MB of 8-byte-per-lane coalesced loads) to calibrate the counter for this
library's access width, then the bench query (persistent sweep) so its traffic can be read per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
eng = Engine(0)
eng.generate_synthetic(n)
exact = make_query(nat.M_EXACT, 100.0)
clt = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4)
blk = make_query(nat.M_BLOCK, 1.0, block_size=1000)
for _ in range(20):
    eng.reduce(exact)
for _ in range(20):
    eng.reduce(clt)
for _ in range(20):
    eng.reduce(blk)
grp = make_query(nat.M_ROWID_MOD, 10.0, agg=nat.AVG)
for _ in range(20):
    eng.reduce_grouped(grp, nat.GROUP_PRODUCT)
print("done")
