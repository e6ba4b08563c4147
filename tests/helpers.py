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
    if m == "direct_access_sample":
        return o.idx_direct_access(N, pct)
    if m == "optimized_sequential_sample":
        return o.idx_optimized_sequential(N, pct, int(a[0]))
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
        # (max_error_percent = 0 never converges: the multiset of rows does not depend on how often the rules are looked
        #  at, and at 10 M rows the reference's cadence would be 100 000 rounds — a coarse schedule gives the same rows)
        sched = dict(R0=4096, growth=4) if (a[3] == 0.0 and N >= 5_000_000) else {}
        rc, res, idx = o.clt_run(rows, pct, a[0], int(a[1]), int(a[2]), a[3], want_idx=True, **sched)
        assert rc == 0
        return idx
    raise KeyError(m)


def rel(a, b):
    return abs(a - b) / max(abs(a), abs(b), 1e-300)


# ---- AQE_M_RANDOM_DEVICE: the keyed bijection of [0, N), restated in numpy from its definition in include/aqe_hip.h /
#      csrc/planner.hpp (PermSpec).  It has no counterpart in the reference (whose random samplers are seeded from
#      std::random_device): this restatement pins the device kernels' index set bit for bit; the statistical tests pin
#      that the set behaves like the reference's shuffled prefix (sample_records, DB.cpp:345-363).
_C1, _C2, _C3 = 0x9E3779B97F4A7C15, 0xBF58476D1CE4E5B9, 0x94D049BB133111EB
_M64 = (1 << 64) - 1


def _splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def perm_rows(N, pct, seed):
    """Rows (draw order) of AQE_M_RANDOM_DEVICE over a table (or row window) of N rows."""
    target = min(int(N * pct / 100.0), N)
    if N == 0 or target <= 0:
        return np.zeros(0, dtype=np.uint64)
    bits = max(1, int(N - 1).bit_length())
    mask = np.uint64((1 << bits) - 1)
    k0, k1 = np.uint64(_splitmix64(seed & _M64)), np.uint64(_splitmix64((seed ^ 0xA5A5A5A5A5A5A5A5) & _M64))
    s1, s2, s3 = (np.uint64(max(1, v)) for v in (bits // 2, bits // 3, (2 * bits) // 3))
    x = np.arange(target, dtype=np.uint64)
    todo = np.ones(target, dtype=bool)
    with np.errstate(over="ignore"):
        while todo.any():
            y = x[todo]
            y = (y + k0) & mask
            y = (y * np.uint64(_C1)) & mask; y ^= y >> s1
            y = (y + k1) & mask
            y = (y * np.uint64(_C2)) & mask; y ^= y >> s2
            y = (y * np.uint64(_C3)) & mask; y ^= y >> s3
            y = (y * np.uint64(_C1)) & mask; y ^= y >> s1
            x[todo] = y
            again = todo.copy()
            again[todo] = y >= np.uint64(N)
            todo = again
    return x


def va_table(oracle_obj, n, lo, hi, ties=False):
    """Rows [lo, hi) of the table the sharded variance-aware sampler tests use: zones of different spread (so
    adaptive_block_sample's block sizes differ between zones) and, with `ties`, amounts rounded to whole numbers (hundreds of
    rows per value: the tie rule of the global sorted positions is exercised)."""
    import numpy as np
    rows = oracle_obj.synth(hi - lo, 42, first=lo)
    i = np.arange(lo, hi)
    scale = 0.2 + 0.8 * ((i * 10 // max(n, 1)) % 3) / 2.0
    amt = 500.5 + (rows["amount"] - 500.5) * scale
    rows["amount"] = np.round(amt) if ties else amt
    return rows


VARIANCE_AWARE = [  # (sampler, pct, block_size / min_block_size, strata / max_block_size)
    ("adaptive", 5.0, 500, 2000),
    ("adaptive", 1.0, 64, 700),
    ("stratified", 2.0, 100, 7),
    ("stratified", 10.0, 1000, 4),
    ("stratified", 0.5, 37, 10),
]


def lognormal_table(oracle_obj):
    """(fixture, rows) of tests/golden/clt_lognormal.json: the skewed table regenerated from numpy's PCG64 stream.  Skips the
    calling test when the platform's exp() rounds differently from where the fixture was recorded (the digest says so)."""
    import hashlib
    import json
    from pathlib import Path

    import numpy as np
    import pytest
    g = json.loads((Path(__file__).parent / "golden" / "clt_lognormal.json").read_text())
    rows = oracle_obj.synth(g["rows"], g["seed"])
    rows["amount"] = np.random.Generator(np.random.PCG64(g["rng_seed"])).lognormal(g["mu"], g["sigma"], g["rows"])
    if hashlib.sha256(np.ascontiguousarray(rows["amount"]).tobytes()).hexdigest() != g["amount_sha256"]:
        pytest.skip("numpy/libm on this platform generate a different log-normal table than the fixture was recorded on")
    return g, rows
