import os, sys, statistics
sys.path.insert(0, "/root/repo")
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
eng = Engine(0); eng.generate_synthetic(10_000_000)
st = torch.cuda.Stream().cuda_stream
for T in (4, 16, 32, 64, 128, 256):
    q = make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4, num_threads=T)
    p = eng.plan(q)
    for _ in range(5):
        p.enqueue_all(st); r = p.fetch(st)
    p.set_profiling(True); ms = []
    for _ in range(100):
        p.enqueue_all(st); r = p.fetch(st); torch.cuda.synchronize(); ms.append(sum(p.launch_ms()))
    print("T", T, "rounds", r.rounds, "n", r.n, "kernel", nat.KERNEL_NAMES[p.last_kernel()], "launches", len(p.launch_ms()), "us %.2f" % (1e3 * statistics.median(ms)))
    p.close()
