"""Diagnostics (GPU box): in-kernel timeline of k_round from a library built with -DAQE_ROUND_STAMPS
(tools/ab_libs.sh rstamps "-DAQE_ROUND_STAMPS"; AQE_HIP_LIB=tools/lib_rstamps.bin python tools/stamp_round.py [exact|s20|s1|b20] [rows])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
what = sys.argv[1] if len(sys.argv) > 1 else "exact"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
eng = Engine(0)
eng.generate_synthetic(rows)
q = {"exact": make_query(nat.M_EXACT, 100.0), "s20": make_query(nat.M_MEMORY_STRIDE, 20.0), "s1": make_query(nat.M_MEMORY_STRIDE, 1.0),
     "b20": make_query(nat.M_BLOCK, 20.0, block_size=1000)}[what]
st = torch.cuda.Stream().cuda_stream
p = eng.plan(q)
lib = nat.lib()
lib.aqe_debug_round_stamps.argtypes = [C.c_void_p, C.c_size_t]
W = 2048 * 4
buf = np.zeros((W + 1) * 8, dtype=np.uint64)
zero = np.zeros_like(buf)
for it in range(8):
    p.enqueue_all(st); r = p.fetch(st)
    torch.cuda.synchronize()
    lib.aqe_debug_round_stamps(buf.ctypes.data, buf.size)
    if it < 3:
        continue
    w = buf[: W * 8].reshape(W, 8).astype(np.int64)
    f = buf[W * 8:].astype(np.int64)
    t0 = w[w[:, 0] > 0, 0].min()
    # only waves of THIS launch: entry within 100 us of the earliest
    live = (w[:, 0] >= t0) & (w[:, 0] < t0 + 100000)
    us = lambda x: (x - t0) / 100.0
    col = lambda k: w[live & (w[:, k] >= t0), k]
    print("%s %dM (%d waves): starts ..%.2f | table staged %.2f..%.2f | first tile %.2f..%.2f | sweep done %.2f..%.2f | workgroup summed ..%.2f | partial out %.2f..%.2f | ticket %.2f..%.2f | fold: partials read %.2f published %.2f" % (
        what, rows // 1000000, int(live.sum()), us(col(0).max()), us(col(1).min()), us(col(1).max()), us(col(2).min()), us(col(2).max()), us(col(3).min()), us(col(3).max()),
        us(col(4).max()), us(col(5).min()), us(col(5).max()), us(col(6).min()), us(col(6).max()), us(f[0]), us(f[1])))
