"""Dev helper (GPU box): device time of the dense sweeps under the library AQE_HIP_LIB names (A/B of load policies)."""
import os, sys, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Batch, Engine, make_query
import bench
eng = Engine(0)
out = [os.path.basename(os.environ.get("AQE_HIP_LIB", "default"))]
def med(q, reps=15):
    for _ in range(3): eng.reduce(q)
    return statistics.median(eng.reduce(q).kernel_ms for _ in range(reps)) * 1e3
for n, tag in ((100_000_000, "100M"), (1_000_000_000, "1B")):
    eng.generate_synthetic(n)
    out.append(f"{tag} exact {med(make_query(nat.M_EXACT, 100.0)):.1f}")
    out.append(f"stride20 {med(make_query(nat.M_MEMORY_STRIDE, 20.0)):.1f}")
    out.append(f"block20 {med(make_query(nat.M_BLOCK, 20.0)):.1f}")
    if n == 10_000_000:
        st = torch.cuda.Stream().cuda_stream
        ps = [eng.plan(q) for q in bench.headline_queries(nat, make_query, 32, 1, 0.01)]
        b = Batch(ps)
        for _ in range(5): b.enqueue_all(st); b.fetch()
        b.set_profiling(True)
        ms = []
        for _ in range(30): b.enqueue_all(st); b.fetch(); ms.append(b.launch_info()[0])
        out.append(f"batch32 {1e3 * statistics.median(ms):.1f}")
        b.close()
        for p in ps: p.close()
print(" | ".join(out))
