"""Dev helper (GPU box): launch time of CLT and strided queries on SMALL tables (100 k / 1 M rows) under the library AQE_HIP_LIB names."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
st = torch.cuda.Stream().cuda_stream
out = [os.path.basename(os.environ.get("AQE_HIP_LIB", "default"))]
for rows in (100_000, 1_000_000):
    eng = Engine(0)
    eng.generate_synthetic(rows)
    for name, q in (("clt e=0", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.0, clt_round0=1024, clt_growth=4)),
                    ("clt e=1%", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=1.0, clt_round0=1024, clt_growth=4)),
                    ("s20", make_query(nat.M_MEMORY_STRIDE, 20.0)), ("s5", make_query(nat.M_MEMORY_STRIDE, 5.0))):
        p = eng.plan(q)
        for _ in range(10):
            p.enqueue_all(st); p.fetch(st)
        p.set_profiling(True)
        ms = []
        for _ in range(200):
            p.enqueue_all(st); r = p.fetch(st)
            torch.cuda.synchronize()
            ms.append(sum(p.launch_ms()))
        p.set_profiling(False)
        out.append("%dk %s %.2f (%s, %d launches)" % (rows // 1000, name, 1e3 * statistics.median(ms), nat.KERNEL_NAMES[p.last_kernel()], len(p.launch_ms())))
        p.close()
    eng.close()
print(" | ".join(out))
