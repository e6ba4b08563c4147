"""Dev helper: soak the one-launch batch (mixed kinds: k_sweep_multi; `clt` as second argument: CLT plans only, k_sweep_lean_multi) — three batches of mixed queries (CLT never converging / stopping
early with the top-up / stopping in the middle, strided, block + WHERE, exact, pages) alternate on two streams, a
single-plan launch runs in between on a third, every result of every step is fetched and compared BITWISE with the
first execution of its plan.  A protocol failure shows up as AqeError (device_status) or a changed answer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Batch, Engine, make_query
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
eng = Engine(0)
eng.generate_synthetic(10_000_000)
def clt(e, r0, g, t, agg=nat.AVG):
    return make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=agg, max_error_percent=e, clt_round0=r0, clt_growth=g, num_threads=t)
kinds = [clt(0.01, 4096, 4, 4), clt(1.0, 4096, 4, 6, nat.SUM), clt(0.05, 512, 2, 8), clt(0.0, 64, 3, 6, nat.COUNT), clt(0.03, 1024, 4, 10),
         make_query(nat.M_MEMORY_STRIDE, 1.0), make_query(nat.M_MEMORY_STRIDE, 20.0, agg=nat.AVG), make_query(nat.M_BLOCK, 1.0, where=(250.0, 750.0), convention=nat.EST_CPP),
         make_query(nat.M_EXACT, 100.0), make_query(nat.M_PAGE, 5.0, block_size=4096)]
if len(sys.argv) > 2 and sys.argv[2] == "clt":  # only plans that qualify for the lean groups (k_sweep_lean_multi)
    kinds = kinds[:5] + [clt(0.5, 256, 2, 12), clt(2.0, 4096, 4, 16, nat.SUM)]
sizes = (32, 7, 64)
batches, plans = [], []
for n in sizes:
    ps = [eng.plan(kinds[i % len(kinds)]) for i in range(n)]
    plans.append(ps)
    batches.append(Batch(ps))
solo = eng.plan(kinds[0])
streams = [torch.cuda.Stream() for _ in range(3)]
key = lambda r: (r.n, r.visited, r.converged, r.rounds, r.topup, r.sum, r.sumsq, r.value, r.ci_lower, r.device_status, r.topup_pending)
first = [None] * len(batches)
solo_first = None
t0 = time.perf_counter()
done = 0
for it in range(iters):
    k = it % len(batches)
    batches[k].enqueue_all(streams[k % 2].cuda_stream)
    if it % 5 == 0:
        solo.enqueue_all(streams[2].cuda_stream)
        s = key(solo.fetch(streams[2].cuda_stream))
        solo_first = solo_first or s
        assert s == solo_first, (it, s, solo_first)
    j = (it - 1) % len(batches)
    if it:
        got = [key(r) for r in batches[j].fetch()]
        if first[j] is None:
            first[j] = got
        assert got == first[j], (it, j, [(a, b) for a, b in zip(got, first[j]) if a != b][:1])
        done += len(got)
got = [key(r) for r in batches[(iters - 1) % len(batches)].fetch()]
dt = time.perf_counter() - t0
print(f"soak_batch ok: {done} batched queries + {iters // 5} single launches in {dt:.1f} s; bitwise equal to their first execution; "
      f"first batch: {[(g[0], g[2], g[3], g[4]) for g in first[0][:5]]}")
