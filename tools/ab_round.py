"""Dev helper (GPU box): single-round sampler launch times under the library AQE_HIP_LIB names."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
eng = Engine(0)
eng.generate_synthetic(10_000_000)
out = []
for name, q in (("stride1", make_query(nat.M_MEMORY_STRIDE, 1.0)), ("block1", make_query(nat.M_BLOCK, 1.0)), ("stride20", make_query(nat.M_MEMORY_STRIDE, 20.0)),
                ("exact", make_query(nat.M_EXACT, 100.0))):
    p = eng.plan(q)
    p.set_profiling(True)
    acc = []
    for _ in range(200):
        p.enqueue_all()
        p.fetch()
        acc.append(sum(p.launch_ms()))
    acc.sort()
    out.append(f"{name} {1e3 * acc[len(acc) // 2]:.2f}")
print(os.path.basename(os.environ.get("AQE_HIP_LIB", "default")), " ".join(out))
