"""Test helpers: map a golden "call" (reference method + positional args) onto the oracle."""
import zlib

import numpy as np


def digest(idx):
    idx = np.ascontiguousarray(idx, dtype="<u8")
    return {
        "n": int(len(idx)),
        "first": [int(x) for x in idx[:16]],
        "last": [int(x) for x in idx[-16:]],
        "crc32": zlib.crc32(idx.tobytes()) & 0xFFFFFFFF,
        "wsum": int(int(idx.sum(dtype=np.uint64)) % (1 << 64)),
    }


def oracle_indices(o, rows, call, cache_rows=None):
    """Row indices the oracle predicts for one reference call.  cache_rows = length of the reference's
    flat cache (floor(N/1000)*1000 after plain inserts, DB.cpp:188-191) for the two cached samplers."""
    N = len(rows)
    M = N if cache_rows is None else cache_rows
    m, pct, a = call["method"], call["pct"], call["args"]
    if m == "memory_stride_sample":
        return o.idx_memory_stride(M, pct, int(a[0]))
    if m == "optimized_address_arithmetic_sample":
        return o.idx_address_arithmetic(M, pct)
    if m == "random_pointer_sample":
        return o.idx_random_pointer(N, pct, int(a[0]))
    if m == "block_sample":
        return o.idx_block(N, pct, int(a[0]))
    if m == "page_sample":
        return o.idx_page(N, pct, int(a[0]))
    if m == "parallel_block_sample":
        return o.idx_parallel_block(N, pct, int(a[0]), int(a[1]))
    if m == "optimized_clt_sample":
        return o.idx_optimized_clt(N, pct, int(a[2]))
    if m == "fast_pointer_sample":
        return o.idx_fast_pointer(N, pct, int(a[0]))
    if m == "slow_pointer_sample":
        return o.idx_slow_pointer(N, pct)
    if m == "dual_pointer_sample":
        return o.idx_dual_pointer(N, pct)
    if m == "parallel_pointer_sample":
        return o.idx_parallel_pointer(N, pct, int(a[0]))
    if m == "adaptive_block_sample":
        return o.idx_adaptive_block(rows, pct, int(a[0]), int(a[1]))
    if m == "stratified_block_sample":
        return o.idx_stratified_block(rows, pct, int(a[0]), int(a[1]))
    if m == "clt_validated_dual_pointer_sample":
        rc, res, idx = o.clt_run(rows, pct, a[0], int(a[1]), int(a[2]), a[3], want_idx=True)
        assert rc == 0
        return idx
    raise KeyError(m)


def rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-300)
