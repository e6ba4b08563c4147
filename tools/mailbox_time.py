"""tools/mailbox_time.py — what one peer-mapped all-reduce (aqe_mailbox_all_reduce_sum) costs on THIS box: G contexts of one
process on one GPU (peers connected directly, so no xGMI hop is in the number: it is the floor — launch, stores, flags, sum),
one launch per rank on its own stream, host clock around a round of G launches + synchronize and the device's own begin-to-end
time of rank 0's launch.      python tools/mailbox_time.py [G ...]   ->  profiles/round3_mailbox.txt (via tools/gpu_final3.sh)"""
import statistics
import sys
import time

import torch

sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parent.parent))
from approximatequeryengine_amd import _native as nat  # noqa: E402
from approximatequeryengine_amd.engine import Engine, Mailbox  # noqa: E402


def main():
    gs = [int(a) for a in sys.argv[1:]] or [1, 2, 4]  # (one process has 4 hardware queues: more ranks than that on ONE device queue up behind each other's spinning launches)
    for G in gs:
        engs = [Engine(0) for _ in range(G)]
        for e in engs:
            e.generate_synthetic(1024, seed=42, keep_aos=False)
        mbs = [Mailbox(e, G, g) for g, e in enumerate(engs)]
        Mailbox.connect_local(mbs)
        streams = [torch.cuda.Stream() for _ in range(G)]
        for count in (8, 40, 1280):
            vecs = [torch.ones(count, dtype=torch.float64, device="cuda") for _ in range(G)]
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            wall, dev = [], []
            for it in range(230):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for g in range(G):
                    if g == 0:
                        ev0.record(streams[0])
                    mbs[g].all_reduce_sum(vecs[g].data_ptr(), count, streams[g].cuda_stream)
                    if g == 0:
                        ev1.record(streams[0])
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                if it >= 30:
                    wall.append(1e6 * (t1 - t0))
                    dev.append(1e3 * ev0.elapsed_time(ev1))
                for v in vecs:
                    v.fill_(1.0)
            late = [m.late_ranks() for m in mbs]
            print(f"ranks {G} doubles {count:5d}: rank 0's launch begin->end median {statistics.median(dev):7.2f} us (min {min(dev):6.2f}); "
                  f"host: {G} launches + synchronize median {statistics.median(wall):7.2f} us; late {late}", flush=True)
        for e in engs:
            e.close()


if __name__ == "__main__":
    main()
