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

constexpr unsigned kRegBins = 8;

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
};

// One wave folds one tile of the family table (the decomposition add_family made: kDenseTileOrdinals for dense
// families, kTileOrdinals otherwise; PAIR families do not occur in single-round samplers).
__global__ __launch_bounds__(kBlockThreads) void k_grouped(GroupLaunch a) {
    // [4][nbins], component-major: the lanes of a wave mostly hold consecutive keys (product_id = row % 100), and
    // consecutive f64 words spread over all LDS banks — [nbins][4] put them 32 bytes apart, on a quarter of the banks
    extern __shared__ double bins[];
    __shared__ DevFamily lds_fams[kMaxLdsFams];
    for (unsigned i = threadIdx.x; i < a.nbins * 4; i += kBlockThreads) bins[i] = 0.0;
    const DevFamily* fams = stage_families(a.sw, lds_fams);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const u64 wave_id = uniform64(static_cast<u64>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6));
    const u64 wave_stride = static_cast<u64>(gridDim.x) * kWavesPerBlock;
    const bool in_regs = a.nbins <= kRegBins;
    const unsigned nb = a.nbins;  // uniform: bins past it cost nothing
    double rs[kRegBins], rq[kRegBins];
    unsigned cn[kRegBins], cv[kRegBins];
#pragma unroll
    for (unsigned b = 0; b < kRegBins; ++b) { rs[b] = rq[b] = 0.0; cn[b] = cv[b] = 0u; }

    for (u64 t = wave_id; t < a.ntiles; t += wave_stride) {
        unsigned lo = 0, hi = a.sw.nfam;
        while (hi - lo > 1) {
            unsigned mid = (lo + hi) >> 1;
            if (fams[mid].tile_begin <= t) lo = mid; else hi = mid;
        }
        const DevFamily& F = fams[lo];
        const u64 lt = t - F.tile_begin;
        u64 seg, j;
        if (F.tiles_per_seg == 0) { seg = F.seg_lo; j = F.j_lo + lt; }
        else { seg = F.seg_lo + lt / F.tiles_per_seg; j = lt % F.tiles_per_seg; }
        const u64 seg_len = F.seg_len, step = F.step, seg_ord0 = seg * seg_len;
        const u64 tile = (a.sw.dense16 && is_dense16(step, F.flags, seg_len)) ? kDenseTileOrdinals : kTileOrdinals;
        const u64 row_base = F.row0 + seg * F.pitch - a.sw.shard_lo;
        for (u64 k0 = 0; k0 < tile; k0 += static_cast<u64>(64) * kTileUnroll) {
            double x[kTileUnroll];
            int key[kTileUnroll];
            bool ok[kTileUnroll];
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {  // every load of the batch is issued before the first use
                const u64 oi = j * tile + k0 + static_cast<u64>(k) * 64 + lane;
                const u64 o = seg_ord0 + oi;
                ok[k] = oi < seg_len && o >= F.ord_lo && o < F.ord_hi;
                const u64 row = ok[k] ? row_base + oi * step : 0;
                x[k] = a.sw.amount[row];
                key[k] = a.keys[row];
            }
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {
                const bool pass = ok[k] && (!a.sw.has_where || (x[k] >= a.sw.wmin && x[k] <= a.sw.wmax));  // DB.cpp:329
                const double d = x[k] - a.sw.shift;
                const unsigned b = static_cast<unsigned>(key[k] - a.key_min);
                if (in_regs) {
#pragma unroll
                    for (unsigned g = 0; g < kRegBins; ++g) {
                        if (g < nb) {
                            const bool mine = ok[k] && b == g, counted = mine && pass;
                            cv[g] += mine ? 1u : 0u;
                            cn[g] += counted ? 1u : 0u;
                            rs[g] += counted ? d : 0.0;
                            rq[g] += counted ? d * d : 0.0;
                        }
                    }
                } else if (ok[k] && b < a.nbins) {
                    atomicAdd(&bins[3 * nb + b], 1.0);
                    if (pass) {
                        atomicAdd(&bins[b], 1.0);
                        atomicAdd(&bins[nb + b], d);
                        atomicAdd(&bins[2 * nb + b], d * d);
                    }
                }
            }
        }
    }
    if (in_regs) {
        // the wave's totals, seven values at a time (value i = bins[i]: bin i / 4, component i % 4); lane 8 c of a
        // batch holds its c-th total and adds it to the workgroup's bin
        constexpr unsigned kVals = kRegBins * 4;
#pragma unroll
        for (unsigned i0 = 0; i0 < kVals; i0 += 7) {
            if (i0 < nb * 4) {
                double v[7];
#pragma unroll
                for (unsigned c7 = 0; c7 < 7; ++c7) {
                    const unsigned i = i0 + c7, g = i / 4, comp = i % 4;  // compile-time after unrolling
                    v[c7] = i >= kVals ? 0.0 : comp == 0 ? static_cast<double>(cn[g]) : comp == 1 ? rs[g] : comp == 2 ? rq[g] : static_cast<double>(cv[g]);
                }
                const double total = wave_sum7(v, lane);
                const unsigned i = i0 + (static_cast<unsigned>(lane) >> 3);
                if ((lane & 7) == 0 && lane < 56 && i < nb * 4 && total != 0.0) atomicAdd(&bins[(i % 4) * nb + i / 4], total);
            }
        }
    }
    __syncthreads();
    double* out = a.partial + static_cast<size_t>(blockIdx.x) * a.nbins * 4;
    for (unsigned i = threadIdx.x; i < a.nbins * 4; i += kBlockThreads) out[i] = bins[(i % 4) * nb + i / 4];  // out: [nbins][4]
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
                          unsigned grid, hipStream_t s) {
    GroupLaunch a{sw, ntiles, keys, key_min, nbins, partial};
    hipLaunchKernelGGL(k_grouped, dim3(grid), dim3(kBlockThreads), nbins * 4 * sizeof(double), s, a);
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
