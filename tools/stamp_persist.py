"""Diagnostics: run the bench query a few times with AQE_PERSIST_STAMPS set and print the in-kernel timeline."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.environ.setdefault("AQE_PERSIST_STAMPS", "/tmp/aqe_stamps.txt")
if os.path.exists(out):
    os.remove(out)
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
e = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
eng = Engine(0)
eng.generate_synthetic(10_000_000)
q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=e, clt_round0=4096, clt_growth=4)
# argv[2] = "auto": the form the library picks (the head form for a query predicted to stop early);
# default: the timeline of the full single-launch form
q.flags = 0 if (len(sys.argv) > 2 and sys.argv[2] == "auto") else nat.Q_FORCE_PERSIST
for _ in range(6):
    r = eng.reduce(q)
print(r.n, r.rounds, r.converged, r.kernel_ms)
print(open(out).read())
