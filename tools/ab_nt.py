"""Dev helper (GPU box): the load policy of the lean sweep — plain against non-temporal loads (AQE_NT=0 / 1, read per plan) — by
bytes swept, around the 256 MiB Infinity Cache where the library switches.     python tools/ab_nt.py"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query

st = torch.cuda.Stream().cuda_stream
eng = Engine(0)
for rows in [int(a) for a in sys.argv[1:]] or (20_000_000, 40_000_000, 60_000_000, 100_000_000, 200_000_000):
    eng.generate_synthetic(rows)
    cases = [("exact", make_query(nat.M_EXACT, 100.0)), ("stride 20%", make_query(nat.M_MEMORY_STRIDE, 20.0)), ("block 20%", make_query(nat.M_BLOCK, 20.0)),
             ("CLT e=0.01%", make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4))]
    for name, q in cases:
        line = []
        for mode in (("1", "0", "1", "0") if os.environ.get("AB_ORDER") == "swap" else ("0", "1", "0", "1")):
            os.environ["AQE_NT"] = mode
            p = eng.plan(q)
            for _ in range(5):
                p.enqueue_all(st); r = p.fetch(st)
            p.set_profiling(True)
            us = []
            own = torch.cuda.Stream() if os.environ.get("AB_FRESH") == "1" else None  # (bench.py times a case on a stream of its own)
            ts = own.cuda_stream if own is not None else st
            if os.environ.get("AB_SYNC") == "1":
                torch.cuda.synchronize()
            for _ in range(30):
                p.enqueue_all(ts); r = p.fetch(ts)
                us.append(1e3 * sum(p.launch_ms()))
            p.set_profiling(False)
            med = statistics.median(us)
            line.append("nt=%s %.2f us frac %.3f" % (mode, med, 8.0 * r.visited / (med * 1e-6) / 8e12))
            p.close()
        print("%4dM %-12s %7.0f MB | %s" % (rows // 1_000_000, name, 8e-6 * r.visited, " | ".join(line)), flush=True)
os.environ.pop("AQE_NT", None)
eng.close()
