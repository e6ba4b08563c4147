// grouped.hip — GROUP BY with a per-group interval: the sampled sweep with one (n, S, Q) bin per key.
//
// Reference: execute_query_groupby_with_ci, src/aqe_backend/executor.cpp:202-321 (the reference's SQLite path:
// one SQL statement per distinct key, each a full pass over the table).  Here ONE sweep of the sampled rows reads
// the amount and the key of each row and bins the shifted moments:
//
//   * keys are dense small integers (region 0..3, product_id 0..99): bin = key - key_min, at most kMaxGroupBins;
//   * a workgroup accumulates into LDS-privatised bins.  With up to kRegBins bins (region) every lane keeps its own
//     bins in registers (counts as integers), the wave sums them with VALU cross-lane moves (wave_sum7) and ONE lane
//     per value adds the wave's total to LDS — 64 lanes hammering 4 LDS addresses with atomics serialise (the
//     first version did that in its epilogue: 23 us for a 100 k-row sample); with more bins (product_id) lanes add
//     straight to LDS (ds_add_f64), where collisions are rare;
//   * workgroups write their bins to a [workgroup][bin][4] buffer, k_grouped_sum adds them per bin in workgroup
//     order (this is what ranks all-reduce in the multi-GPU form: the bins are additive), and k_grouped_finish
//     works out mean, variance, estimate and interval (executor.cpp:277-296).
//
// Counts are exact.  Within a workgroup the LDS additions happen in whatever order the lanes arrive, so the
// floating-point sums of a group are reproducible to rounding (1e-15 relative), not bit for bit.
#include "device_common.hpp"

namespace aqe {
namespace {

// Binning.  The bins of a workgroup live in LDS; how depends on the number of keys:
//   * up to kPrivBins keys (region: 4): LANE-PRIVATE bins, [component][bin][thread].  Few keys mean every lane of a
//     wave instruction hits one of a handful of addresses — shared bins serialise (64 lanes on 4 addresses), and
//     per-lane register bins need a masked add per bin and element (what this kernel did before: 73 us for the exact
//     scan of 10 M rows, bound by those adds).  With a word of its own per lane and bin, an LDS add never conflicts,
//     costs the same for every key, and the workgroup's totals are summed afterwards in thread order: bit-reproducible.
//   * more keys (product_id: 100): bins shared by the workgroup, component-major, lanes add with ds_add_f64 — into one
//     of up to 8 REPLICAS of the bins, picked by the lane (lane & 7), each replica an odd number of words long.  The
//     lanes of a wave instruction hold rows 2 L or 2 L + 1 (16-byte loads): with keys that follow the row number
//     (product_id = row % 100) that is a 16-byte stride in one copy of the bins — 8 lanes on each of 8 bank pairs,
//     measured 34 us for the exact scan of 10 M rows, bound by the LDS — and an even spread over all banks with the
//     replicas; lanes that hold the same key mostly add to different words.
constexpr unsigned kPrivBins = 8;
constexpr unsigned kMaxReplicas = 8;
constexpr unsigned kSharedLdsBytes = 48u << 10;
__host__ __device__ inline unsigned replica_stride(unsigned nbins) { return nbins | 1u; }  // odd: replicas start on different banks
__host__ __device__ inline unsigned replicas_for(unsigned nbins) {
    unsigned r = kSharedLdsBytes / (replica_stride(nbins) * 4u * 8u);
    r = r > kMaxReplicas ? kMaxReplicas : r;
    unsigned p = 1;
    while (2 * p <= r) p *= 2;  // a power of two (lane & (p - 1))
    return p;
}

template <bool kPrivate>
struct Binner {
    double* s;      // sum of (x - c)
    double* q;      // sum of (x - c)^2
    unsigned* n;    // rows that pass WHERE            (private mode: u32 counters; shared mode: f64, see below)
    unsigned* v;    // rows sampled
    double* nd;     // shared mode: counts as f64 (one LDS atomic type)
    double* vd;
    unsigned nb, tid;
    unsigned rep_off;  // shared mode: this lane's replica, as an offset into each component's array
    __device__ __forceinline__ void add(unsigned b, bool ok, bool pass, double d) const {
        if (kPrivate) {
            const unsigned i = b * kBlockThreads + tid;
            if (ok) {
                __hip_atomic_fetch_add(v + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (pass) {
                    __hip_atomic_fetch_add(n + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(s + i, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    __hip_atomic_fetch_add(q + i, d * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
            }
        } else if (ok && b < nb) {
            const unsigned i = rep_off + b;
            __hip_atomic_fetch_add(vd + i, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (pass) {
                __hip_atomic_fetch_add(nd + i, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(s + i, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                __hip_atomic_fetch_add(q + i, d * d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
    }
};


__global__ __launch_bounds__(kBlockThreads) void k_extract_key(const aqe_record* __restrict__ aos, int32_t* __restrict__ out, u64 n,
                                                               int column) {
    for (u64 i = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; i < n; i += static_cast<u64>(gridDim.x) * kBlockThreads)
        out[i] = column == AQE_GROUP_REGION ? aos[i].region : aos[i].product_id;
}

// keys of the synthetic `sales` table (kernels.hip k_synth): region = i % 4, product_id = i % 100
__global__ __launch_bounds__(kBlockThreads) void k_synth_key(int32_t* __restrict__ out, u64 n, u64 first_row, int column) {
    const u64 m = column == AQE_GROUP_REGION ? 4 : 100;
    for (u64 k = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; k < n; k += static_cast<u64>(gridDim.x) * kBlockThreads)
        out[k] = static_cast<int32_t>((first_row + k) % m);
}

// min and max key of the column: out[0] = min, out[1] = max (initialised by the host to INT_MAX / INT_MIN)
__global__ __launch_bounds__(kBlockThreads) void k_key_range(const int32_t* __restrict__ keys, u64 n, int32_t* out) {
    int lo = 2147483647, hi = -2147483647 - 1;
    for (u64 i = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; i < n; i += static_cast<u64>(gridDim.x) * kBlockThreads) {
        const int k = keys[i];
        lo = k < lo ? k : lo;
        hi = k > hi ? k : hi;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const int l2 = __shfl_xor(lo, off, 64), h2 = __shfl_xor(hi, off, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    // one pair of atomics per workgroup (a same-address device atomic costs ~20 ns and serialises)
    __shared__ int wlo[kWavesPerBlock], whi[kWavesPerBlock];
    if ((threadIdx.x & 63) == 0) { wlo[threadIdx.x >> 6] = lo; whi[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < kWavesPerBlock; ++w) { lo = wlo[w] < lo ? wlo[w] : lo; hi = whi[w] > hi ? whi[w] : hi; }
        atomicMin(&out[0], lo);
        atomicMax(&out[1], hi);
    }
}

struct GroupLaunch {
    SweepCommon sw;
    u64 ntiles;
    const int32_t* keys;  // this shard's key column
    int32_t key_min;
    uint32_t nbins;
    double* partial;      // [gridDim.x][nbins][4]: n, S - c n, Q (shifted), visited
    GroupFuse fuse;       // kFused: the launch also adds the workgroups' bins up and works every group out (single GPU)
};

// Estimate and interval of one group from its sums (executor.cpp:277-296).
__device__ __forceinline__ aqe_group_result group_result(double n, double sd, double qd, double visited, int64_t key, double shift, double pct, int agg) {
    aqe_group_result r;
    r.key = key;
    r.n = static_cast<uint64_t>(n);
    r.visited = static_cast<uint64_t>(visited);
    const double c = shift;
    r.sum = sd + n * c;
    r.sumsq = qd + 2.0 * c * sd + n * c * c;
    double mean = 0.0, m2 = 0.0;
    if (n > 0.0) mean_m2(n, sd, qd, c, mean, m2);
    r.mean = mean;
    const double scale = 100.0 / pct;
    double margin = 0.0;
    if (n >= 2.0) margin = 1.96 * sqrt((m2 / (n - 1.0)) / n);  // executor.cpp:280-286
    double value;
    if (agg == AQE_SUM) { value = r.sum * scale; margin *= scale; }  // executor.cpp:289-296 scales the interval so
    else if (agg == AQE_AVG) { value = mean; }
    else { value = n * scale; margin = 0.0; }
    r.value = value;
    r.ci_lower = value - margin;
    r.ci_upper = value + margin;
    return r;
}

// The fused epilogue (single GPU, aqe_reduce_grouped): every workgroup ADDS its bins to one [nbins][4] accumulator in
// device memory (f64 atomic adds, executed at the memory side; counts are integers and stay exact, the sums were
// arrival-ordered within a workgroup already), drains, draws a ticket (sharded, as k_round's); the workgroup that draws
// the last one reads the accumulator, clears it for the next launch, works every group out and writes it — with a check
// word over its fields — into the pinned block the host polls.  One launch and no stream synchronisation, where the sweep,
// a second launch over the workgroups' bins and a wait for the stream took 35-60 us per call for a 9-28 us sweep.
__device__ __forceinline__ void grouped_fused_epilogue(const GroupLaunch& a, const double* mine /* LDS: this workgroup's [nbins][4] */, double* lds_tot) {
    __shared__ int s_last;
    const unsigned nb = a.nbins, words = nb * 4;
    for (unsigned i = threadIdx.x; i < words; i += kBlockThreads) {
        const double v = mine[i];
        if (v != 0.0) unsafeAtomicAdd(a.fuse.acc + i, v);  // (global_atomic_add_f64: device memory, no compare-and-swap loop)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this thread's adds have been performed ...
    __syncthreads();                                    // ... and so have the workgroup's, before its ticket is drawn
    if (threadIdx.x == 0) {
        const unsigned G = gridDim.x, shards = G < static_cast<unsigned>(kShards) ? G : static_cast<unsigned>(kShards);
        unsigned* const ct = a.fuse.ticket + static_cast<size_t>(kShards) * kShardStride;
        int last = 0;
        if (G <= 16u) {
            if (__hip_atomic_fetch_add(ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == G - 1u) { __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = 1; }
        } else {
            const unsigned sh = blockIdx.x % shards, members = (G - sh + shards - 1u) / shards;
            unsigned* const cs = a.fuse.ticket + static_cast<size_t>(sh) * kShardStride;
            if (__hip_atomic_fetch_add(cs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
                __hip_atomic_store(cs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__hip_atomic_fetch_add(ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1u) { __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); last = 1; }
            }
        }
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    for (unsigned i = threadIdx.x; i < words; i += kBlockThreads) {
        lds_tot[i] = __hip_atomic_load(a.fuse.acc + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.fuse.acc + i, 0.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the next launch starts from zero)
    }
    __syncthreads();
    for (unsigned b = threadIdx.x; b < nb; b += kBlockThreads) {
        const aqe_group_result r = group_result(lds_tot[b * 4 + 0], lds_tot[b * 4 + 1], lds_tot[b * 4 + 2], lds_tot[b * 4 + 3], static_cast<int64_t>(a.key_min) + b,
                                                a.fuse.shift, a.fuse.pct, a.fuse.agg);
        a.fuse.out[b] = r;
        __hip_atomic_store(a.fuse.check + b, group_check(r, a.fuse.epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One wave folds one tile of the family table (the decomposition add_family made: kDenseTileOrdinals for dense
// families — two rows per lane per load: 16 bytes of amounts, 8 bytes of keys — kTileOrdinals otherwise; PAIR families
// do not occur in single-round samplers).
template <bool kPrivate, bool kFused>
__global__ __launch_bounds__(kBlockThreads) void k_grouped(GroupLaunch a) {
    extern __shared__ double lds[];
    __shared__ DevFamily lds_fams[kMaxLdsFams];
    const unsigned nb = a.nbins;
    const unsigned reps = replicas_for(nb), rstride = replica_stride(nb), comp_len = reps * rstride;
    const unsigned words = kPrivate ? nb * kBlockThreads * 3 : comp_len * 4;  // in doubles (private: s, q, and n + v as 2 x u32)
    for (unsigned i = threadIdx.x; i < words; i += kBlockThreads) lds[i] = 0.0;
    Binner<kPrivate> B;
    B.nb = nb;
    B.tid = threadIdx.x;
    if (kPrivate) {
        B.s = lds;
        B.q = lds + nb * kBlockThreads;
        B.n = reinterpret_cast<unsigned*>(lds + 2 * nb * kBlockThreads);
        B.v = B.n + nb * kBlockThreads;
        B.nd = B.vd = nullptr;
    } else {
        B.nd = lds; B.s = lds + comp_len; B.q = lds + 2 * comp_len; B.vd = lds + 3 * comp_len;
        B.n = B.v = nullptr;
        B.rep_off = (threadIdx.x & (reps - 1u)) * rstride;
    }
    const DevFamily* fams = stage_families(a.sw, lds_fams);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u64 wave_id = uniform64(static_cast<u64>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6));
    const u64 wave_stride = static_cast<u64>(gridDim.x) * kWavesPerBlock;
    const double c = a.sw.shift, wmin = a.sw.wmin, wmax = a.sw.wmax;
    const bool has_where = a.sw.has_where != 0;
    const int kmin = a.key_min;

    for (u64 t = wave_id; t < a.ntiles; t += wave_stride) {
        const DevFamily& F = fams[find_family(fams, a.sw.nfam, t)];
        const u64 lt = t - F.tile_begin;
        u64 seg, j;
        if (F.tiles_per_seg == 0) { seg = F.seg_lo; j = F.j_lo + lt; }
        else { seg = F.seg_lo + lt / F.tiles_per_seg; j = lt % F.tiles_per_seg; }
        const u64 seg_len = F.seg_len, step = F.step, seg_ord0 = seg * seg_len;
        const u64 ord_lo = F.ord_lo, ord_hi = F.ord_hi;
        const u64 row_base = F.row0 + seg * F.pitch - a.sw.shard_lo;
        const double* const base = a.sw.amount + row_base;
        const int32_t* const kbase = a.keys + row_base;
        if (a.sw.dense16 && is_dense16(step, F.flags, seg_len)) {
            struct __attribute__((packed, aligned(8))) Row2 { double x, y; };
            struct __attribute__((packed, aligned(4))) Key2 { int x, y; };
            const u64 oi0 = j * kDenseTileOrdinals + 2 * static_cast<u64>(lane);
            Row2 x2[kTileUnroll];
            Key2 k2[kTileUnroll];
            bool ok0[kTileUnroll], ok1[kTileUnroll];
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {  // every load of the tile is issued before the first use
                const u64 oi = oi0 + static_cast<u64>(k) * 128;
                const u64 o = seg_ord0 + oi;
                ok0[k] = oi < seg_len && o >= ord_lo && o < ord_hi;
                ok1[k] = oi + 1 < seg_len && o + 1 >= ord_lo && o + 1 < ord_hi;
                const bool both = ok0[k] && ok1[k];
                x2[k] = *reinterpret_cast<const Row2*>(both ? base + oi : a.sw.amount);
                k2[k] = *reinterpret_cast<const Key2*>(both ? kbase + oi : a.keys);
                if (!both) {  // window edge: single reads
                    x2[k].x = ok0[k] ? base[oi] : 0.0;
                    x2[k].y = ok1[k] ? base[oi + 1] : 0.0;
                    k2[k].x = ok0[k] ? kbase[oi] : kmin;
                    k2[k].y = ok1[k] ? kbase[oi + 1] : kmin;
                }
            }
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {
                const bool p0 = !has_where || (x2[k].x >= wmin && x2[k].x <= wmax);  // inclusive both ends, DB.cpp:329
                const bool p1 = !has_where || (x2[k].y >= wmin && x2[k].y <= wmax);
                B.add(static_cast<unsigned>(k2[k].x - kmin), ok0[k], p0, x2[k].x - c);
                B.add(static_cast<unsigned>(k2[k].y - kmin), ok1[k], p1, x2[k].y - c);
            }
            continue;
        }
        const u64 oi0 = j * kTileOrdinals + lane;
        double x[kTileUnroll];
        int key[kTileUnroll];
        bool ok[kTileUnroll];
        if (F.flags & kFamLinear) {
            // short segments (pages) tiled along the ordinal axis, a tile spanning several segments (device_common.hpp, sweep_family)
            const u64 T0 = j * kTileOrdinals;
            const u64 seg0 = T0 / seg_len;
            const unsigned r0 = static_cast<unsigned>(T0 - seg0 * seg_len), sl = static_cast<unsigned>(seg_len);
            const float inv = 1.0f / static_cast<float>(sl);
            const u64 col0 = F.row0 - a.sw.shard_lo;
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {
                const unsigned xx = r0 + static_cast<unsigned>(lane) + 64u * static_cast<unsigned>(k);
                const unsigned qx = static_cast<unsigned>((static_cast<float>(xx) + 0.5f) * inv);
                const u64 o = T0 + static_cast<unsigned>(lane) + 64u * static_cast<unsigned>(k);
                ok[k] = o >= ord_lo && o < ord_hi;
                const u64 off = ok[k] ? col0 + (seg0 + qx) * F.pitch + static_cast<u64>(xx - qx * sl) * step : 0;
                x[k] = a.sw.amount[off];
                key[k] = ok[k] ? a.keys[off] : kmin;
            }
        } else {
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {
                const u64 oi = oi0 + static_cast<u64>(k) * 64;
                const u64 o = seg_ord0 + oi;
                ok[k] = oi < seg_len && o >= ord_lo && o < ord_hi;
                const u64 off = ok[k] ? oi * step : 0;
                x[k] = ok[k] ? base[off] : a.sw.amount[0];
                key[k] = ok[k] ? kbase[off] : kmin;
            }
        }
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) {
            const bool pass = !has_where || (x[k] >= wmin && x[k] <= wmax);
            B.add(static_cast<unsigned>(key[k] - kmin), ok[k], pass, x[k] - c);
        }
    }
    __syncthreads();
    // [nbins][4]: n, S - c n, Q, visited — into the workgroup's slice of the partial buffer, or (fused) into LDS behind the bins
    double* out = kFused ? lds + words : a.partial + static_cast<size_t>(blockIdx.x) * nb * 4;
    if (kPrivate) {
        // the workgroup's 256 private words per (bin, component), summed by a fixed binary tree over the threads
        for (unsigned stride = kBlockThreads / 2; stride > 0; stride >>= 1) {
            if (threadIdx.x < stride) {
                for (unsigned b = 0; b < nb; ++b) {
                    const unsigned i = b * kBlockThreads + threadIdx.x;
                    B.s[i] += B.s[i + stride];
                    B.q[i] += B.q[i + stride];
                    B.n[i] += B.n[i + stride];
                    B.v[i] += B.v[i + stride];
                }
            }
            __syncthreads();
        }
        if (threadIdx.x < nb * 4) {
            const unsigned b = threadIdx.x >> 2, comp = threadIdx.x & 3, w = b * kBlockThreads;
            out[threadIdx.x] = comp == 0 ? static_cast<double>(B.n[w]) : comp == 1 ? B.s[w] : comp == 2 ? B.q[w] : static_cast<double>(B.v[w]);
        }
    } else {
        for (unsigned i = threadIdx.x; i < nb * 4; i += kBlockThreads) {  // out: [nbins][4]; the replicas in order
            const unsigned comp = i % 4, b = i / 4;
            double t = 0.0;
            for (unsigned r = 0; r < reps; ++r) t += lds[comp * comp_len + r * rstride + b];
            out[i] = t;
        }
    }
    if (kFused) {
        __syncthreads();
        grouped_fused_epilogue(a, out, out + nb * 4);
    }
}

// One wave per (bin, component): lane l adds the workgroups l, l + 64, ... in order, then a fixed xor butterfly
// adds the lanes -> bins[nbins][4].  (One thread per bin walking a thousand workgroups' partials one dependent
// load after the other took 100 us; this takes a few.)
__global__ __launch_bounds__(64) void k_grouped_sum(const double* __restrict__ partial, unsigned nblocks, unsigned nbins, double* __restrict__ bins) {
    const unsigned i = blockIdx.x;  // (bin, component)
    const unsigned lane = threadIdx.x;
    double t = 0.0;
    for (unsigned w = lane; w < nblocks; w += 64) t += partial[static_cast<size_t>(w) * nbins * 4 + i];
    for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off, 64);
    if (lane == 0) bins[i] = t;
}

// One thread per bin: estimate and interval of the group from its (all-reduced) sums.
__global__ __launch_bounds__(64) void k_grouped_finish(const double* __restrict__ bins, unsigned nbins, int32_t key_min, double shift, double pct,
                                                       int agg, aqe_group_result* __restrict__ out) {
    const unsigned b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nbins) return;
    out[b] = group_result(bins[b * 4 + 0], bins[b * 4 + 1], bins[b * 4 + 2], bins[b * 4 + 3], static_cast<int64_t>(key_min) + b, shift, pct, agg);
}

// Single-GPU form: sum and finish in one launch.  One wave per bin: lane l adds the workgroups l, l + 64, ... of each
// of the bin's four components (the same order and butterfly as k_grouped_sum), then lane 0 works the group out.
__global__ __launch_bounds__(64) void k_grouped_sum_finish(const double* __restrict__ partial, unsigned nblocks, unsigned nbins, int32_t key_min,
                                                           double shift, double pct, int agg, aqe_group_result* __restrict__ out) {
    const unsigned b = blockIdx.x, lane = threadIdx.x;
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (unsigned w = lane; w < nblocks; w += 64) {
        const double* p = partial + (static_cast<size_t>(w) * nbins + b) * 4;
#pragma unroll
        for (int cmp = 0; cmp < 4; ++cmp) t[cmp] += p[cmp];
    }
#pragma unroll
    for (int cmp = 0; cmp < 4; ++cmp)
        for (int off = 32; off > 0; off >>= 1) t[cmp] += __shfl_xor(t[cmp], off, 64);
    if (lane == 0) out[b] = group_result(t[0], t[1], t[2], t[3], static_cast<int64_t>(key_min) + b, shift, pct, agg);
}

inline unsigned blocks_for(u64 work, u64 per_block, unsigned cap) {
    u64 g = (work + per_block - 1) / per_block;
    if (g < 1) g = 1;
    return static_cast<unsigned>(g > cap ? cap : g);
}

}  // namespace

hipError_t launch_extract_key(const aqe_record* aos, int32_t* out, uint64_t n, int column, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_extract_key, dim3(blocks_for(n, kBlockThreads * 4, kMaxBlocks)), dim3(kBlockThreads), 0, s, aos, out, static_cast<u64>(n), column);
    return hipGetLastError();
}

hipError_t launch_synth_key(int32_t* out, uint64_t n, uint64_t first_row, int column, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth_key, dim3(blocks_for(n, kBlockThreads * 4, kMaxBlocks)), dim3(kBlockThreads), 0, s, out, static_cast<u64>(n),
                       static_cast<u64>(first_row), column);
    return hipGetLastError();
}

hipError_t launch_key_range(const int32_t* keys, uint64_t n, int32_t* out2, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_key_range, dim3(blocks_for(n, kBlockThreads * 16, 512)), dim3(kBlockThreads), 0, s, keys, static_cast<u64>(n), out2);
    return hipGetLastError();
}

unsigned grouped_grid(uint64_t ntiles) { return blocks_for(ntiles, kWavesPerBlock, kGroupedMaxBlocks); }

hipError_t launch_grouped(const SweepCommon& sw, uint64_t ntiles, const int32_t* keys, int32_t key_min, uint32_t nbins, double* partial,
                          unsigned grid, hipStream_t s, const GroupFuse* fuse) {
    GroupLaunch a{sw, ntiles, keys, key_min, nbins, partial, fuse ? *fuse : GroupFuse{}};
    const size_t bins_bytes = nbins <= kPrivBins ? static_cast<size_t>(nbins) * kBlockThreads * 3 * sizeof(double)
                                                 : static_cast<size_t>(replicas_for(nbins)) * replica_stride(nbins) * 4 * sizeof(double);
    const size_t fuse_bytes = fuse ? static_cast<size_t>(nbins) * 8 * sizeof(double) : 0;  // the workgroup's [nbins][4] + the folded [nbins][4]
    if (nbins <= kPrivBins) {
        if (fuse) hipLaunchKernelGGL((k_grouped<true, true>), dim3(grid), dim3(kBlockThreads), bins_bytes + fuse_bytes, s, a);
        else hipLaunchKernelGGL((k_grouped<true, false>), dim3(grid), dim3(kBlockThreads), bins_bytes, s, a);
    } else {
        if (fuse) hipLaunchKernelGGL((k_grouped<false, true>), dim3(grid), dim3(kBlockThreads), bins_bytes + fuse_bytes, s, a);
        else hipLaunchKernelGGL((k_grouped<false, false>), dim3(grid), dim3(kBlockThreads), bins_bytes, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_grouped_sum(const double* partial, unsigned nblocks, uint32_t nbins, double* bins, hipStream_t s) {
    hipLaunchKernelGGL(k_grouped_sum, dim3(nbins * 4), dim3(64), 0, s, partial, nblocks, nbins, bins);
    return hipGetLastError();
}

hipError_t launch_grouped_sum_finish(const double* partial, unsigned nblocks, uint32_t nbins, int32_t key_min, double shift, double pct, int agg,
                                     aqe_group_result* out, hipStream_t s) {
    hipLaunchKernelGGL(k_grouped_sum_finish, dim3(nbins), dim3(64), 0, s, partial, nblocks, nbins, key_min, shift, pct, agg, out);
    return hipGetLastError();
}

hipError_t launch_grouped_finish(const double* bins, uint32_t nbins, int32_t key_min, double shift, double pct, int agg, aqe_group_result* out,
                                 hipStream_t s) {
    hipLaunchKernelGGL(k_grouped_finish, dim3((nbins + 63) / 64), dim3(64), 0, s, bins, nbins, key_min, shift, pct, agg, out);
    return hipGetLastError();
}

}  // namespace aqe
