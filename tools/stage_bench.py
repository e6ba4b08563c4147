"""Dev helper: host -> HBM staging rate of aqe_stage_records / aqe_stage_file (PCIe-inclusive, never part of `value`)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from approximatequeryengine_amd.engine import Engine
from oracle.pyoracle import Oracle
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50_000_000
o = Oracle()
t0 = time.perf_counter(); rows = o.synth(n, 42); print(f"synth {n} rows {time.perf_counter()-t0:.2f}s", flush=True)
eng = Engine(0)
for keep in (False, True):
    for rep in range(2):
        t0 = time.perf_counter(); eng.stage_records(rows, keep_aos=keep); dt = time.perf_counter() - t0
        moved = n * (32 if keep else 8)
        print(f"stage_records keep_aos={keep}: {dt*1e3:.1f} ms  host rows {n*32/dt/1e9:.2f} GB/s read, {moved/dt/1e9:.2f} GB/s over PCIe", flush=True)
path = os.path.join(tempfile.gettempdir(), "stage_bench.db")
o.file_write(path, rows)
for keep in (False, True):
    t0 = time.perf_counter(); eng.stage_file(path, keep_aos=keep); dt = time.perf_counter() - t0
    print(f"stage_file keep_aos={keep}: {dt*1e3:.1f} ms  {n*32/dt/1e9:.2f} GB/s of file", flush=True)
os.remove(path)
