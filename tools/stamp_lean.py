"""Diagnostics (GPU box): in-kernel timeline of k_sweep_lean from a library built with -DAQE_LEAN_STAMPS
(tools/ab_libs.sh stamps "-DAQE_LEAN_STAMPS"; AQE_HIP_LIB=tools/lib_stamps.bin python tools/stamp_lean.py [clt|exact|s20])."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from approximatequeryengine_amd import _native as nat
from approximatequeryengine_amd.engine import Engine, make_query
what = sys.argv[1] if len(sys.argv) > 1 else "clt"
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
eng = Engine(0)
eng.generate_synthetic(rows)
q = {"clt": make_query(nat.M_CLT_DUAL_POINTER, 20.0, agg=nat.AVG, max_error_percent=0.01, clt_round0=4096, clt_growth=4),
     "exact": make_query(nat.M_EXACT, 100.0), "s20": make_query(nat.M_MEMORY_STRIDE, 20.0)}[what]
st = torch.cuda.Stream().cuda_stream
p = eng.plan(q)
lib = nat.lib()
lib.aqe_debug_lean_stamps.argtypes = [C.c_void_p, C.c_size_t]
W = 256 * 16
buf = np.zeros((W + 1) * 8, dtype=np.uint64)
for it in range(8):
    p.enqueue_all(st); r = p.fetch(st)
    torch.cuda.synchronize()
    assert p.last_kernel() == nat.KERNEL_SWEEP_LEAN, p.last_kernel()
    lib.aqe_debug_lean_stamps(buf.ctypes.data, buf.size)
    if it < 3:
        continue
    w = buf[: W * 8].reshape(W, 8).astype(np.int64)
    f = buf[W * 8:].astype(np.int64)
    live = w[:, 0] > 0
    t0 = w[live, 0].min()
    us = lambda x: (x - t0) / 100.0
    col = lambda k: w[live & (w[:, k] > 0), k]
    print("%s %dM: starts ..%.2f | table %.2f..%.2f | first tile %.2f..%.2f | sweep done %.2f..%.2f | handed ..%.2f | partial out %.2f..%.2f | ticket %.2f..%.2f | fold: rounds summed %.2f judged %.2f" % (
        what, rows // 1000000, us(col(0).max()), us(col(1).min()), us(col(1).max()), us(col(2).min()), us(col(2).max()), us(col(3).min()), us(col(3).max()), us(col(4).max()),
        us(col(5).min()), us(col(5).max()), us(col(6).min()), us(col(6).max()), us(f[1]), us(f[2])))
