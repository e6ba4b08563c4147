"""Dev helper (GPU box): the load policy A/B of tools/ab_nt.py, but measured the way bench.py measures a case (measure_config:
warm-up, 20 profiled launches on a stream of their own, median) — plain, non-temporal, plain, non-temporal on one box.
    python tools/ab_nt_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query

st = torch.cuda.Stream().cuda_stream
eng = Engine(0)
for rows in (10_000_000, 100_000_000):
    eng.generate_synthetic(rows, seed=42, keep_aos=False)
    for name, q in (("exact", make_query(nat.M_EXACT, 100.0)), ("stride 20%", make_query(nat.M_MEMORY_STRIDE, 20.0)), ("block 20%", make_query(nat.M_BLOCK, 20.0)),
                    ("CLT e=0.01%", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4))):
        line = []
        for mode in ("0", "1", "0", "1"):
            os.environ["AQE_NT"] = mode
            c = bench.measure_config(eng, st, f"{rows // 1_000_000}M {name} nt={mode}", q, reps=20)
            line.append("nt=%s %.2f us (min %.2f) %.3f" % (mode, c["kernel_us"], c["kernel_us_min"], c["frac"]))
        print("%4dM %-12s | %s" % (rows // 1_000_000, name, " | ".join(line)), flush=True)
os.environ.pop("AQE_NT", None)
eng.close()
