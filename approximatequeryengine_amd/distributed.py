"""distributed.py — one process per GPU; the table is sharded by contiguous row region and every
convergence step costs exactly one all-reduce of the moment vector (n, Σ(x-c), Σ(x-c)²)×{fast, slow}.

The reference merges its workers through a mutex-guarded vector, a CAS loop on atomic<double> and an
atomic<bool> stop flag (custom_bplus_db.cpp:948-951, 966-967, 2031-2036).  Here each rank sweeps the
part of every worker's progression that falls inside its shard (the families are clipped on the host,
planner.cpp), RCCL sums the AQE_MOMENT_VEC doubles over xGMI, and every rank folds the same reduced
vector with the same device kernel — so every rank takes the same stop decision and no broadcast of
should_stop is needed.  A round enqueued after the stop is a device-side no-op on every rank.

This module is pure orchestration: it never touches the numbers.  The object it drives only has to
look like ``engine.Plan`` (rounds, has_topup, reset, enqueue_round, enqueue_update, enqueue_finalize,
fetch), which is how the CPU ``gloo`` tests exercise it with an oracle-backed stand-in.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

MOMENT_VEC = 8


def shard_bounds(n_global: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Proper prefix partition of the flat row array: rank g owns [⌊g·N/G⌋, ⌊(g+1)·N/G⌋).
    (The reference's own region split, custom_bplus_db.cpp:1903-1921, overlaps by a row and can drop
    the last N mod T rows; that defect is not reproduced.)"""
    return (rank * n_global) // world_size, ((rank + 1) * n_global) // world_size


class ShardedQuery:
    """Runs one planned query across the ranks of a process group.

    plan        an object with the ``engine.Plan`` interface, planned over THIS rank's shard
    vec         a tensor of MOMENT_VEC float64 on the plan's device (reused every round)
    all_reduce  callable(tensor) -> None performing an in-place SUM over the group
    stream      raw stream handle passed through to the plan (0 = the plan's own stream)
    sync_every  fetch the device stop flag every this many rounds to stop enqueueing early
                (0 = never look: rounds after the stop are cheap device-side no-ops)
    """

    def __init__(self, plan, vec, all_reduce: Callable, stream: int = 0):
        self.plan = plan
        self.vec = vec
        self.all_reduce = all_reduce
        self.stream = stream

    def enqueue(self) -> None:
        p, v, s = self.plan, self.vec, self.stream
        p.reset(s)
        steps = p.rounds + (1 if p.has_topup else 0)
        for r in range(steps):
            v.zero_()                            # a launch that leaves early writes nothing
            p.enqueue_round(r, v.data_ptr(), s)  # this shard's partial (n, Σd, Σd²)
            self.all_reduce(v)                   # ONE collective per convergence step
            p.enqueue_update(r, v.data_ptr(), s) # fold + CLT rules + should_stop, on the device
        p.enqueue_finalize(s)

    def run(self):
        self.enqueue()
        return self.plan.fetch(self.stream)


def torch_all_reduce(group=None) -> Callable:
    """SUM all-reduce through torch.distributed (backend "nccl" is RCCL on ROCm; "gloo" on CPU)."""
    import torch.distributed as dist

    def _ar(t):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)

    return _ar
