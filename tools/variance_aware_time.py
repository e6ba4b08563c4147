"""tools/variance_aware_time.py — the two samplers of SURVEY 8f rank 4 that need a pass over the whole table first:
adaptive_block_sample (ten zone variances: ten exact window scans) and stratified_block_sample (the column sorted by amount:
rocPRIM radix sort + row permutation).  First call (pre-pass included) and steady state, kernel time and fraction of 8 TB/s.
    python tools/variance_aware_time.py [rows ...]        ->  profiles/round3_variance_aware.txt"""
import os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query

st = torch.cuda.Stream().cuda_stream
for rows in [int(a) for a in sys.argv[1:]] or (10_000_000, 100_000_000):
    for name, q in (("adaptive_block 5% (500..2000)", make_query(nat.M_ADAPTIVE_BLOCK, 5.0, block_size=500, block_size_max=2000)),
                    ("stratified_block 5% (B=1000, 4 strata)", make_query(nat.M_STRATIFIED_BLOCK, 5.0, block_size=1000, num_threads=4)),
                    ("stratified_block 20% (B=1000, 10 strata)", make_query(nat.M_STRATIFIED_BLOCK, 20.0, block_size=1000, num_threads=10)),
                    ("block 5% (B=1000) for comparison", make_query(nat.M_BLOCK, 5.0))):
        with Engine(0) as eng:  # a fresh table each time: the pre-pass is paid by the first call
            eng.generate_synthetic(rows, seed=42, keep_aos=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = eng.reduce(q)
            first_ms = 1e3 * (time.perf_counter() - t0)
            p = eng.plan(q)
            for _ in range(3):
                p.enqueue_all(st); p.fetch(st)
            p.set_profiling(True)
            us, lat = [], []
            for _ in range(20):
                p.enqueue_all(st); r = p.fetch(st)
                us.append(1e3 * sum(p.launch_ms()))
            p.set_profiling(False)
            for _ in range(20):
                t1 = time.perf_counter(); p.enqueue_all(st); r = p.fetch(st); lat.append(1e6 * (time.perf_counter() - t1))
            med = statistics.median(us)
            print("%4dM %-42s first call %8.2f ms | then kernel %7.2f us, closed loop %7.2f us | %8d rows, %6.1f MB, %.3f of 8 TB/s | %s" % (
                rows // 1_000_000, name, first_ms, med, statistics.median(lat), r.visited, 8e-6 * r.visited, 8.0 * r.visited / (med * 1e-6) / 8e12,
                nat.KERNEL_NAMES.get(p.last_kernel())), flush=True)
            p.close()
