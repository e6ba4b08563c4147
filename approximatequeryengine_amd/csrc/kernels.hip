// kernels.hip — hand-written gfx950 (CDNA4, wave64) kernels of the sampled-reduce path.
//
// The path is a bandwidth/latency-bound f64 reduction (3 flops per 8 bytes): no MFMA.  What matters is
// coalesced loads of the SoA `amount` column, many loads in flight per lane, a wave64 butterfly, an LDS
// cross-wave step, and ONE hand-off per workgroup to the last-arriving workgroup, which folds the launch
// into the query's running moments and evaluates the CLT rules on the device.
//
//   k_round / k_indexed   one launch = one round, one single-round sampler, or the top-up
//                         (reducer loops DB.cpp:285-294, 324-335, 2024-2036; should_stop DB.cpp:930, 987 is
//                         QueryState::stop tested on kernel entry; top-up DB.cpp:1031-1040)
//   k_update / k_replay   fold all-reduced vectors on every rank (multi-GPU forms)
//   k_finalize            estimators + interval (enhanced_aqe_cli.py:189-200, 277-291; DB.cpp:303-315)
//   k_gather*             record-returning samplers; k_id_bounds key range -> row window
//   k_split_amount, k_synth   staging
// (DB.cpp = /root/reference/src/aqe_backend/core/custom_bplus_db.cpp; the shared device code with the
// rule and estimator restatements is device_common.hpp; the single-launch multi-round sweep is persist.hip.)
#include <hip/hip_ext.h>

#include "device_common.hpp"

namespace aqe {
namespace {

// Diagnostics (builds with -DAQE_ROUND_STAMPS only; tools/stamp_round.py): s_memrealtime marks of k_round, 100 MHz.
// [wave][8]: 0 entry, 1 family table staged, 2 first tile folded, 3 sweep done, 4 workgroup summed, 5 partial out, 6 ticket drawn;
// then [8] of the folding workgroup: 0 partials read, 1 folded and published
#ifdef AQE_ROUND_STAMPS
__device__ unsigned long long g_round_stamps[(kMaxBlocks * kWavesPerBlock + 1) * 8];
#define ROUND_STAMP(slot) do { if ((threadIdx.x & 63) == 0) g_round_stamps[(static_cast<size_t>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6)) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define ROUND_STAMP_FOLD(slot) do { if (threadIdx.x == 0) g_round_stamps[static_cast<size_t>(kMaxBlocks) * kWavesPerBlock * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define ROUND_STAMP(slot) do { } while (0)
#define ROUND_STAMP_FOLD(slot) do { } while (0)
#endif

// Sum 7 per-thread values over the workgroup in a fixed order: wave64 butterfly (wave_sum7), then the
// four wave results through LDS.  The totals are valid in thread 0.  `red` must be quiescent on entry.
__device__ __forceinline__ void block_sum7(double (&v)[7], double (*red)[kVec]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const double mine = wave_sum7(v, lane);
    if ((lane & 7) == 0 && lane < 56) red[wave][lane >> 3] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            double t = red[0][k];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) t += red[w][k];
            v[k] = t;
        }
    }
}

// Device-clock timing of a query (RoundLaunch::want_ticks): workgroup 0 — dispatched first — notes the clock when the
// query's FIRST launch starts, in a spare word behind the arrival counters; the workgroup that folds that launch (the
// last to arrive: workgroup 0's store is long out) carries it into the state, and whoever finishes the query reads
// the clock again.
__device__ __forceinline__ u64* t0_word(const RoundLaunch& a) { return reinterpret_cast<u64*>(a.counter + kShards * kShardStride + 8); }
__device__ __forceinline__ void note_start(const RoundLaunch& a) {
    if (a.want_ticks && a.reset_state && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_store(t0_word(a), static_cast<u64>(__builtin_amdgcn_s_memrealtime()), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The launch that finishes a query writes the result and, for a host that polls the pinned result instead of waiting
// for the end of the launch, the check word beside it (kernels.hpp, result_check).
__device__ __forceinline__ void publish_result(const QueryState& st, const RoundLaunch& a) {
    aqe_result r = make_result(st, a.fin);
    // (timed on the device — 10 ns ticks — so that the host can pick the result up by polling instead of waiting for
    // two event records around the launches to complete)
    if (a.want_ticks) r.kernel_ms = static_cast<double>(__builtin_amdgcn_s_memrealtime() - st.t0) * 1e-5;
    *a.result = r;
    if (a.result_seq) __hip_atomic_store(a.result_seq, result_check(r, a.epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// The same through LDS, for a finishing thread whose whole wave is at hand (finish_block): the thread parks the words,
// lane i of its wave sends word i — ONE store instruction instead of sixteen (a store costs one lane what it costs
// sixty-four: ~25 ns each on the tail of every query).  words[0] says whether there is anything to send.
constexpr unsigned kResWords = sizeof(aqe_result) / 8;
__device__ __forceinline__ void park_result(const QueryState& st, const RoundLaunch& a, unsigned long long* words) {
    aqe_result r = make_result(st, a.fin);
    if (a.want_ticks) r.kernel_ms = static_cast<double>(__builtin_amdgcn_s_memrealtime() - st.t0) * 1e-5;
    *reinterpret_cast<aqe_result*>(words + 1) = r;
    words[1 + kResWords] = result_check(r, a.epoch);
    words[0] = 1ull;
}
__device__ __forceinline__ void send_parked_result(const RoundLaunch& a, const unsigned long long* words) {  // every lane of the parking thread's wave
    const unsigned lane = threadIdx.x & 63u;
    if (words[0] == 0ull) return;
    if (lane < kResWords) __hip_atomic_store(reinterpret_cast<unsigned long long*>(a.result) + lane, words[1 + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (lane == kResWords && a.result_seq) __hip_atomic_store(a.result_seq, words[1 + kResWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Thread 0 of the folding workgroup: write the reduced vector, fold it into the running state
// (starting from zero on a query's first launch), and finish the query on its last launch.
__device__ __forceinline__ void fold_and_finish(const double (&tot)[7], const RoundLaunch& a, unsigned long long* res_words) {
    double vec[kVec];
#pragma unroll
    for (int k = 0; k < 7; ++k) vec[k] = tot[k];
    vec[7] = 0.0;
    res_words[0] = 0ull;
    if (a.out_vec) {
#pragma unroll
        for (int k = 0; k < kVec; ++k) a.out_vec[k] = vec[k];
    }
    if (a.fused) {
        QueryState st;
        if (a.reset_state) {
            st = QueryState{};
            if (a.want_ticks) st.t0 = __hip_atomic_load(t0_word(a), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            st = *a.state;
        }
        fold(st, vec, a.fold);
        // (a query that is ONE launch — first and last at once — leaves no state behind for anybody: fourteen stores saved)
        if (!(a.reset_state && a.do_finalize)) *a.state = st;
        if (a.do_finalize) park_result(st, a, res_words);
    }
}

// Workgroup epilogue shared by k_round and k_indexed: block sum -> one 56-byte partial per workgroup,
// published with write-through (sc1) stores by ONE lane, then an agent-scope ticket.  The workgroup
// that draws the last ticket re-reads every partial with sc1 loads in a fixed order (deterministic
// sum), resets the ticket and folds.  (cdna_hip_programming.md G16, counter form.)
__device__ __forceinline__ void finish_block(const Acc& acc, const RoundLaunch& a) {
    __shared__ double red[kWavesPerBlock][kVec];
    __shared__ int s_last;
    double v[7] = {static_cast<double>(acc.na), acc.sa, acc.qa, static_cast<double>(acc.nb), acc.sb, acc.qb,
                   static_cast<double>(acc.nv)};
    __shared__ unsigned long long res_words[kResWords + 2];
    block_sum7(v, red);
    ROUND_STAMP(4);
    if (gridDim.x == 1) {  // a launch small enough for one workgroup needs no hand-off
        if (threadIdx.x == 0) fold_and_finish(v, a, res_words);
        if (threadIdx.x < 64) { wave_lds_handoff(); send_parked_result(a, res_words); }  // (thread 0's wave)
        return;
    }
    if (threadIdx.x == 0) {
        double* mine = a.partials + static_cast<size_t>(blockIdx.x) * kVec;
#pragma unroll
        for (int k = 0; k < 7; ++k) __hip_atomic_store(mine + k, v[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // partial is out before the ticket is drawn
        ROUND_STAMP(5);
        // two-level ticket: shard (blockIdx % kShards), then the top counter
        const unsigned shards = gridDim.x < static_cast<unsigned>(kShards) ? gridDim.x : static_cast<unsigned>(kShards);
        const unsigned sh = blockIdx.x % shards;
        const unsigned members = (gridDim.x - sh + shards - 1) / shards;
        unsigned* cs = a.counter + static_cast<size_t>(sh) * kShardStride;
        unsigned* ct = a.counter + static_cast<size_t>(kShards) * kShardStride;
        int last = 0;
        if (gridDim.x <= static_cast<unsigned>(kShards)) {  // few workgroups: one counter, one round trip (7.0 against 7.3 us per launch)
            if (__hip_atomic_fetch_add(ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1) {
                __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        } else if (__hip_atomic_fetch_add(cs, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1) {
            __hip_atomic_store(cs, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(ct, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == shards - 1) {
                __hip_atomic_store(ct, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1;
            }
        }
        s_last = last;
        ROUND_STAMP(6);
    }
    __syncthreads();
    if (!s_last) return;
    // Thread t sums the words t, t + 256, ... of the flat partial list — component t & 7 of every 32nd workgroup — with
    // coalesced loads (a wave instruction covers eight whole partials; one lane per workgroup, seven loads each, asked the
    // memory system for every 64-byte partial seven times: 3.3 us for 1024 workgroups, tools/stamp_round.py), sixteen in
    // flight; lanes of equal component then add up over the wave and the waves through LDS, in a fixed order.
    double fs = 0.0;
    const unsigned nwords = gridDim.x * static_cast<unsigned>(kVec);
    for (unsigned w0 = threadIdx.x; w0 < nwords; w0 += 16u * kBlockThreads) {
        double x[16];
#pragma unroll
        for (unsigned i = 0; i < 16u; ++i) {
            const unsigned w = w0 + i * kBlockThreads;
            x[i] = w < nwords ? __hip_atomic_load(a.partials + w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
        }
#pragma unroll
        for (unsigned i = 0; i < 16u; ++i) fs += x[i];
    }
    fs += dpp_f64<0x128>(fs);     // lane ^ 8
    fs = swap_add16(fs, fs);     // lane ^ 16
    fs = swap_add32(fs, fs);     // lane ^ 32: lanes 0..7 (and every lane of their class) hold the wave's sum of component lane & 7
    __syncthreads();  // `red` is reused
    ROUND_STAMP_FOLD(0);
    if ((threadIdx.x & 63) < 8) red[threadIdx.x >> 6][threadIdx.x & 7] = fs;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t[7];
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            double sum = red[0][k];
#pragma unroll
            for (int w = 1; w < kWavesPerBlock; ++w) sum += red[w][k];
            t[k] = sum;
        }
        fold_and_finish(t, a, res_words);
    }
    if (threadIdx.x < 64) { wave_lds_handoff(); send_parked_result(a, res_words); }
    ROUND_STAMP_FOLD(1);
}

// Early-outs every launch of a query shares: should_stop (DB.cpp:930/987) and the top-up gate.
// Returns false when the whole grid must leave without sweeping; the query's last launch still
// finishes the query (one thread) in that case.
__device__ __forceinline__ bool launch_is_live(const RoundLaunch& a, u64& ord_limit) {
    ord_limit = ~0ull;
    bool live = true;
    if (a.fold.is_topup) {
        double collected = a.state->n_p;
        live = collected < static_cast<double>(a.fold.base / 4);  // DB.cpp:1032
        if (live) ord_limit = static_cast<u64>(static_cast<double>(a.fold.base) - collected);  // DB.cpp:1037
    } else if (a.check_stop) {
        live = a.state->stop == 0;
    }
    if (!live && a.fused && a.do_finalize && blockIdx.x == 0 && threadIdx.x == 0) {
        QueryState st = *a.state;
        publish_result(st, a);
    }
    return live;
}

// One wave folds one tile (64 * kTileUnroll ordinals of one segment of one family) per iteration.
template <bool kNT>
__global__ __launch_bounds__(kBlockThreads) void k_round(RoundLaunch a) {
    __shared__ DevFamily lds_fams[kMaxLdsFams];
    u64 ord_limit;
    note_start(a);
    if (!launch_is_live(a, ord_limit)) return;
    const int lane = threadIdx.x & 63;
    const u64 wave_id = uniform64(static_cast<u64>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6));
    const u64 wave_stride = static_cast<u64>(gridDim.x) * kWavesPerBlock;
    Acc acc;
    ROUND_STAMP(0);
    // (the table is staged through LDS also when it has a single entry: passing it in the kernel arguments, as the
    // persistent sweep does, measured 0.5 us SLOWER here — a cold scalar load per wave against one staged copy)
    const DevFamily* fams = stage_families(a.sw, lds_fams);
    ROUND_STAMP(1);
#ifdef AQE_ROUND_STAMPS
    for (u64 t = wave_id; t < a.ntiles; t += wave_stride) { sweep_tile<kNT>(a.sw, fams, t, lane, ord_limit, acc); if (t == wave_id) ROUND_STAMP(2); }
#else
    for (u64 t = wave_id; t < a.ntiles; t += wave_stride) sweep_tile<kNT>(a.sw, fams, t, lane, ord_limit, acc);
#endif
    ROUND_STAMP(3);
    finish_block(acc, a);
}

// random_pointer_sample (DB.cpp:856-882): explicit ascending row list built on the host.
__global__ __launch_bounds__(kBlockThreads) void k_indexed(RoundLaunch a, const uint64_t* __restrict__ idx, u64 n_idx) {
    u64 ord_limit;
    note_start(a);
    if (!launch_is_live(a, ord_limit)) return;
    Acc acc;
    constexpr u64 kChunk = static_cast<u64>(kBlockThreads) * kTileUnroll;
    for (u64 c0 = static_cast<u64>(blockIdx.x) * kChunk; c0 < n_idx; c0 += static_cast<u64>(gridDim.x) * kChunk) {
        u64 row[kTileUnroll];
        bool ok[kTileUnroll];
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) {
            const u64 i = c0 + threadIdx.x + static_cast<u64>(k) * kBlockThreads;
            ok[k] = i < n_idx;
            row[k] = idx[ok[k] ? i : 0];
        }
        double v[kTileUnroll];
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) v[k] = a.sw.amount[row[k] - a.sw.shard_lo];
        TileAcc ta;
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) accumulate(ta, v[k], ok[k], a.sw);
        merge_tile(acc, ta, false);
    }
    finish_block(acc, a);
}

// The keyed bijection of AQE_M_RANDOM_DEVICE (planner.hpp, PermSpec): a row of [0, n) for ordinal k.
__device__ __forceinline__ u64 perm_row(const PermSpec& p, u64 k) {
    u64 x = k;
    do {
        x = (x + p.k0) & p.mask;
        x = (x * kPermC1) & p.mask; x ^= x >> p.s1;
        x = (x + p.k1) & p.mask;
        x = (x * kPermC2) & p.mask; x ^= x >> p.s2;
        x = (x * kPermC3) & p.mask; x ^= x >> p.s3;
        x = (x * kPermC1) & p.mask; x ^= x >> p.s1;
    } while (x >= p.n);  // cycle walking: the bijection of [0, 2^bits) restricted to [0, n); fewer than two turns on average
    return x;
}

// One application of the bijection of [0, 2^bits) (perm_row's loop body).  W = unsigned for tables of up to 2^32 rows:
// every operation is modulo 2^bits, so the low 32 bits of the constants and of the products are all that matters and the
// rows are the same ones, bit for bit — at a quarter of the multiplier work of the 64-bit form.
template <typename W>
__device__ __forceinline__ W perm_round(W x, W k0, W k1, W mask, unsigned s1, unsigned s2, unsigned s3) {
    x = (x + k0) & mask;
    x = (x * static_cast<W>(kPermC1)) & mask; x ^= x >> s1;
    x = (x + k1) & mask;
    x = (x * static_cast<W>(kPermC2)) & mask; x ^= x >> s2;
    x = (x * static_cast<W>(kPermC3)) & mask; x ^= x >> s3;
    x = (x * static_cast<W>(kPermC1)) & mask; x ^= x >> s1;
    return x;
}

// A simple random sample without replacement drawn in the kernel: ordinal k -> row perm.lo + P(k).  No index list to
// build, upload or read (k_indexed reads 8 bytes of index per 8 bytes of amount); a shard keeps the rows it holds.
// Cycle walking diverges: a lane needs 1 / (n / 2^bits) turns per ordinal on average (1.7 on 10 M rows) but the slowest of
// 64 lanes needs six or seven, and walking the eight ordinals of a lane one after the other made every lane wait for the
// slowest lane EIGHT times (11 us for 100 k samples, against 6.6 for the host-built index list).  Each lane now walks its
// own queue of eight ordinals — one turn per wave iteration for whatever ordinal the lane is at — and parks the rows in
// LDS: the wave iterates max over lanes of the SUM of a lane's turns (~20) instead of the sum of the maxima (~55).
template <typename W>
__device__ __forceinline__ void permuted_sweep(const RoundLaunch& a, const PermSpec& perm, u64 shard_lo, u64 shard_rows, Acc& acc) {
    __shared__ unsigned long long lds_row[kTileUnroll][kBlockThreads];
    const W k0 = static_cast<W>(perm.k0), k1 = static_cast<W>(perm.k1), mask = static_cast<W>(perm.mask), n = static_cast<W>(perm.n);
    constexpr u64 kChunk = static_cast<u64>(kBlockThreads) * kTileUnroll;
    for (u64 c0 = static_cast<u64>(blockIdx.x) * kChunk; c0 < perm.target; c0 += static_cast<u64>(gridDim.x) * kChunk) {
        const u64 i0 = c0 + threadIdx.x;  // this lane's ordinals: i0 + k * kBlockThreads
        int k = 0;
        W x = static_cast<W>(i0 < perm.target ? i0 : 0);
        while (__any(k < kTileUnroll)) {
            if (k < kTileUnroll) {
                x = perm_round<W>(x, k0, k1, mask, perm.s1, perm.s2, perm.s3);
                if (x < n) {  // inside [0, n): this ordinal's row
                    lds_row[k][threadIdx.x] = static_cast<unsigned long long>(x);
                    ++k;
                    const u64 i = i0 + static_cast<u64>(k) * kBlockThreads;
                    x = static_cast<W>(i < perm.target ? i : 0);
                }
            }
        }
        double v[kTileUnroll];
        bool ok[kTileUnroll];
#pragma unroll
        for (int j = 0; j < kTileUnroll; ++j) {
            const u64 i = i0 + static_cast<u64>(j) * kBlockThreads;
            const u64 r = perm.lo + lds_row[j][threadIdx.x] - shard_lo;  // (wraps below the shard: fails the test)
            ok[j] = i < perm.target && r < shard_rows;
            v[j] = a.sw.amount[ok[j] ? r : 0];
        }
        TileAcc ta;
#pragma unroll
        for (int j = 0; j < kTileUnroll; ++j) accumulate(ta, v[j], ok[j], a.sw);
        merge_tile(acc, ta, false);
    }
}

__global__ __launch_bounds__(kBlockThreads) void k_permuted(RoundLaunch a, PermSpec perm, u64 shard_lo, u64 shard_rows) {
    u64 ord_limit;
    note_start(a);
    if (!launch_is_live(a, ord_limit)) return;
    Acc acc;
    if (perm.mask <= 0xffffffffull) permuted_sweep<unsigned>(a, perm, shard_lo, shard_rows, acc);
    else permuted_sweep<u64>(a, perm, shard_lo, shard_rows, acc);
    finish_block(acc, a);
}

// record-returning form: out[k] = row perm.lo + P(k) (whole table in this context), in draw order
__global__ __launch_bounds__(kBlockThreads) void k_gather_permuted(const aqe_record* __restrict__ aos, PermSpec perm, aqe_record* __restrict__ out) {
    const uint4* src = reinterpret_cast<const uint4*>(aos);
    uint4* dst = reinterpret_cast<uint4*>(out);
    for (u64 i = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; i < perm.target; i += static_cast<u64>(gridDim.x) * kBlockThreads) {
        const u64 row = perm.lo + perm_row(perm, i);
        uint4 a = src[2 * row], b = src[2 * row + 1];
        dst[2 * i] = a;
        dst[2 * i + 1] = b;
    }
}

__global__ void k_update(QueryState* s, const double* vec, FoldParams p, int reset_state) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    QueryState st;
    if (reset_state) st = QueryState{}; else st = *s;
    if (p.is_topup) {
        if (!(st.n_p < static_cast<double>(p.base / 4))) return;  // same gate as the top-up launch
    } else if (st.stop) {
        return;  // a round enqueued after the stop never ran: nothing to fold
    }
    double v[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) v[k] = vec[k];
    fold(st, v, p);
    *s = st;
}

// Multi-GPU form: `totals` holds the all-reduced total of every round, in order.  Replays the folds in round
// order — lane q keeps the state after round q and judges that round's stop rule, all rounds at once — takes
// the first stopping round (or the last) and writes the state and the result; a due top-up is marked, not run.
__device__ __forceinline__ void replay_one(const double* __restrict__ totals, unsigned rounds, unsigned has_topup, const FoldParams& fp,
                                           const FinalizeParams& fin, QueryState* state, aqe_result* result, int lane) {
    const bool have = static_cast<unsigned>(lane) < rounds;
    double tot_q[7];
#pragma unroll
    for (int c = 0; c < 7; ++c) tot_q[c] = have ? totals[static_cast<size_t>(lane) * kVec + c] : 0.0;
    double run[7] = {0, 0, 0, 0, 0, 0, 0}, mine[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (unsigned q = 0; q < rounds; ++q) {
#pragma unroll
        for (int c = 0; c < 7; ++c) {
            run[c] += read_lane_f64(tot_q[c], static_cast<int>(q));
            if (static_cast<unsigned>(lane) == q) mine[c] = run[c];
        }
    }
    int code = 0;
    if (fp.is_clt && have) code = clt_rules(mine[0], mine[1], mine[2], mine[3], mine[4], mine[5], fp);
    const unsigned long long stops = __ballot(code != 0);
    unsigned last = rounds ? rounds - 1 : 0;
    if (stops) last = static_cast<unsigned>(__builtin_ctzll(stops));
    if (static_cast<unsigned>(lane) != last) return;
    QueryState st{};
    if (rounds) {
        st.n_a = mine[0]; st.sd_a = mine[1]; st.qd_a = mine[2];
        st.n_b = mine[3]; st.sd_b = mine[4]; st.qd_b = mine[5];
        st.n_p = mine[0] + mine[3]; st.sd_p = mine[1] + mine[4]; st.qd_p = mine[2] + mine[5];
        st.visited = mine[6];
        st.rounds = static_cast<int32_t>(last + 1);
        st.converged = code;
        st.stop = code != 0;
    }
    *state = st;
    aqe_result r = make_result(st, fin);
    // DB.cpp:1032: too few rows collected -> the systematic top-up is due.  It is not swept speculatively (a
    // second pass over the table for a rare case): the caller runs it as one more step when it sees the mark.
    r.topup_pending = (has_topup && st.n_p < static_cast<double>(fp.base / 4)) ? 1 : 0;
    *result = r;
}

__global__ void k_replay(const double* __restrict__ totals, unsigned rounds, unsigned has_topup, FoldParams fp,
                         FinalizeParams fin, QueryState* state, aqe_result* result) {
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    replay_one(totals, rounds, has_topup, fp, fin, state, result, threadIdx.x & 63);
}

// A batch of plans after ONE all-reduce: workgroup i replays plan i from row i of the reduced buffer.
__global__ __launch_bounds__(64) void k_replay_batch(const ReplayItem* __restrict__ items, const double* __restrict__ totals, u64 row_stride) {
    const ReplayItem it = items[blockIdx.x];
    replay_one(totals + static_cast<size_t>(blockIdx.x) * row_stride, it.rounds, it.has_topup, it.fp, it.fin, it.state, it.result,
               threadIdx.x & 63);
}

__global__ void k_finalize(const QueryState* s, FinalizeParams p, aqe_result* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    QueryState st = *s;
    finalize(st, p, out);
}

// ---- record-returning samplers: gather 32-byte rows ---------------------------------------------
__global__ __launch_bounds__(kBlockThreads) void k_gather(const aqe_record* __restrict__ aos, u64 shard_lo,
                                                          const DevFamily* __restrict__ fams, unsigned nfam,
                                                          u64 ntiles, aqe_record* __restrict__ out, int dense16,
                                                          const uint32_t* __restrict__ perm) {
    const int lane = threadIdx.x & 63;
    const u64 wave_id = uniform64(static_cast<u64>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6));
    const u64 wave_stride = static_cast<u64>(gridDim.x) * kWavesPerBlock;
    const uint4* src = reinterpret_cast<const uint4*>(aos);
    uint4* dst = reinterpret_cast<uint4*>(out);
    for (u64 t = wave_id; t < ntiles; t += wave_stride) {
        unsigned lo = 0, hi = nfam;
        while (hi - lo > 1) {
            unsigned mid = (lo + hi) >> 1;
            if (fams[mid].tile_begin <= t) lo = mid; else hi = mid;
        }
        const DevFamily& F = fams[lo];
        const u64 lt = t - F.tile_begin;
        u64 seg, j;
        if (F.tiles_per_seg == 0) { seg = F.seg_lo; j = F.j_lo + lt; }
        else { seg = F.seg_lo + lt / F.tiles_per_seg; j = lt % F.tiles_per_seg; }
        const u64 seg_ord0 = seg * F.seg_len;
        const u64 row_base = F.row0 + seg * F.pitch - shard_lo;
        const bool pair = (F.flags & AQE_F_PAIR) != 0;
        const u64 tile = (dense16 && is_dense16(F.step, F.flags, F.seg_len)) ? kDenseTileOrdinals : kTileOrdinals;
        for (u64 half = 0; half * kTileOrdinals < tile; ++half)
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) {
            const u64 oi = j * tile + half * kTileOrdinals + lane + static_cast<u64>(k) * 64;
            const u64 o = seg_ord0 + oi;
            if (oi < F.seg_len && o >= F.ord_lo && o < F.ord_hi) {
                u64 row = row_base + oi * F.step;
                if (perm) row = perm[row];  // position in the amount-sorted table -> row
                const u64 pos = F.out_begin + (o - F.ord_lo);
                uint4 a = src[2 * row], b = src[2 * row + 1];
                dst[2 * pos] = a;
                dst[2 * pos + 1] = b;
            }
            if (pair && o >= F.ord_lo_b && o < F.ord_hi_b) {
                const u64 row = F.row0_b - shard_lo + o * F.step;
                const u64 pos = F.out_begin_b + (o - F.ord_lo_b);
                uint4 a = src[2 * row], b = src[2 * row + 1];
                dst[2 * pos] = a;
                dst[2 * pos + 1] = b;
            }
        }
    }
}

__global__ __launch_bounds__(kBlockThreads) void k_gather_indexed(const aqe_record* __restrict__ aos, u64 shard_lo,
                                                                  const uint64_t* __restrict__ idx, u64 n,
                                                                  aqe_record* __restrict__ out) {
    const uint4* src = reinterpret_cast<const uint4*>(aos);
    uint4* dst = reinterpret_cast<uint4*>(out);
    for (u64 i = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; i < n;
         i += static_cast<u64>(gridDim.x) * kBlockThreads) {
        const u64 row = idx[i] - shard_lo;
        uint4 a = src[2 * row], b = src[2 * row + 1];
        dst[2 * i] = a;
        dst[2 * i + 1] = b;
    }
}

// B+-tree key bounds: first row whose id is >= key[0] and first row whose id is > key[1] (ids ascend in leaf
// order).  Two independent bisections, one lane each.
__global__ void k_id_bounds(const aqe_record* __restrict__ aos, u64 n, long long id_min, long long id_max, u64* out) {
    if (blockIdx.x != 0 || threadIdx.x > 1) return;
    const bool upper = threadIdx.x == 1;
    u64 a = 0, b = n;
    while (a < b) {
        const u64 mid = a + (b - a) / 2;
        const long long id = aos[mid].id;
        const bool go_right = upper ? id <= id_max : id < id_min;
        if (go_right) a = mid + 1; else b = mid;
    }
    out[threadIdx.x] = a;
}

// ---- staging ------------------------------------------------------------------------------------
// AoS -> SoA: each lane reads the 16-byte half-row holding (id, amount) and keeps the amount.
__global__ __launch_bounds__(kBlockThreads) void k_split_amount(const aqe_record* __restrict__ aos,
                                                                double* __restrict__ amount, u64 n) {
    const double2* src = reinterpret_cast<const double2*>(aos);
    for (u64 i = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; i < n;
         i += static_cast<u64>(gridDim.x) * kBlockThreads)
        amount[i] = src[2 * i].y;
}

// Stride-major view of the amount column for step s: row r (global) goes to slot (r % s) * M + (r / s - q0), so
// that the progression row0, row0 + s, row0 + 2 s, ... of a strided pointer is CONTIGUOUS in the view.
__global__ __launch_bounds__(kBlockThreads) void k_stride_view(const double* __restrict__ amount, u64 n, u64 shard_lo, u64 step, u64 M,
                                                               u64 q0, double* __restrict__ out) {
    for (u64 i = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; i < n; i += static_cast<u64>(gridDim.x) * kBlockThreads) {
        const u64 r = shard_lo + i;
        out[(r % step) * M + (r / step - q0)] = amount[i];
    }
}

// ... and a key column in the same slot order (GROUP BY over a strided sample reads amounts and keys side by side)
__global__ __launch_bounds__(kBlockThreads) void k_stride_view_keys(const int32_t* __restrict__ keys, u64 n, u64 shard_lo, u64 step, u64 M,
                                                                    u64 q0, int32_t* __restrict__ out) {
    for (u64 i = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; i < n; i += static_cast<u64>(gridDim.x) * kBlockThreads) {
        const u64 r = shard_lo + i;
        out[(r % step) * M + (r / step - q0)] = keys[i];
    }
}

__device__ __forceinline__ u64 splitmix64_at(u64 seed, u64 i) {
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

// Synthetic `sales` rows: id=i+1, amount=1+999*u, region=i%4, product_id=i%100, timestamp=i.
__global__ __launch_bounds__(kBlockThreads) void k_synth(aqe_record* __restrict__ aos, double* __restrict__ amount,
                                                         u64 n, u64 first_row, u64 seed) {
    for (u64 k = static_cast<u64>(blockIdx.x) * kBlockThreads + threadIdx.x; k < n;
         k += static_cast<u64>(gridDim.x) * kBlockThreads) {
        const u64 i = first_row + k;
        const double u = static_cast<double>(splitmix64_at(seed, i) >> 11) * (1.0 / 9007199254740992.0);
        const double x = 1.0 + 999.0 * u;
        amount[k] = x;
        if (aos) {
            aqe_record r;
            r.id = static_cast<int64_t>(i + 1);
            r.amount = x;
            r.region = static_cast<int32_t>(i % 4);
            r.product_id = static_cast<int32_t>(i % 100);
            r.timestamp = static_cast<int64_t>(i);
            aos[k] = r;
        }
    }
}

inline unsigned grid_for(u64 work_items, u64 per_block) {
    u64 g = (work_items + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > kMaxBlocks) g = kMaxBlocks;
    return static_cast<unsigned>(g);
}

}  // namespace

#ifdef AQE_ROUND_STAMPS
extern "C" __attribute__((visibility("default"))) int aqe_debug_round_stamps(unsigned long long* out, size_t words) {
    const size_t all = (kMaxBlocks * kWavesPerBlock + 1) * 8;
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_round_stamps), 8 * (words < all ? words : all), 0, hipMemcpyDeviceToHost));
}
#endif

hipError_t launch_round(const RoundLaunch& a, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    unsigned grid = grid_for(a.ntiles, kWavesPerBlock);
    // Four workgroups per compute unit, all resident at once (the kernel's registers allow five): 4096 waves hold 32 MB of
    // loads in flight, several times what the memory system needs.  Twice as many workgroups queue for a second turn, and
    // a launch only starts them as the first ones end: 10 M rows exact 25.4 -> 20.5 us, 100 M rows stride 20 % 36.0 -> 32.3 us
    // (A/B on one box, tools/ab_latency.py).
    if (grid > kRoundGridCap) grid = kRoundGridCap;
    // with events: they take the dispatch's own begin/end timestamps (what rocprofv3 reports as the kernel's duration)
    if (a.sw.nt) {
        if (ev0) hipExtLaunchKernelGGL(k_round<true>, dim3(grid), dim3(kBlockThreads), 0, s, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(k_round<true>, dim3(grid), dim3(kBlockThreads), 0, s, a);
    } else {
        if (ev0) hipExtLaunchKernelGGL(k_round<false>, dim3(grid), dim3(kBlockThreads), 0, s, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(k_round<false>, dim3(grid), dim3(kBlockThreads), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_indexed(const RoundLaunch& a, const uint64_t* idx, uint64_t n_idx, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    unsigned grid = grid_for(n_idx, static_cast<u64>(kBlockThreads) * kTileUnroll);
    if (ev0) hipExtLaunchKernelGGL(k_indexed, dim3(grid), dim3(kBlockThreads), 0, s, ev0, ev1, 0, a, idx, static_cast<u64>(n_idx));
    else hipLaunchKernelGGL(k_indexed, dim3(grid), dim3(kBlockThreads), 0, s, a, idx, static_cast<u64>(n_idx));
    return hipGetLastError();
}

hipError_t launch_permuted(const RoundLaunch& a, const PermSpec& perm, uint64_t shard_lo, uint64_t shard_rows, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    unsigned grid = grid_for(perm.target, static_cast<u64>(kBlockThreads) * kTileUnroll);
    if (ev0) hipExtLaunchKernelGGL(k_permuted, dim3(grid), dim3(kBlockThreads), 0, s, ev0, ev1, 0, a, perm, static_cast<u64>(shard_lo), static_cast<u64>(shard_rows));
    else hipLaunchKernelGGL(k_permuted, dim3(grid), dim3(kBlockThreads), 0, s, a, perm, static_cast<u64>(shard_lo), static_cast<u64>(shard_rows));
    return hipGetLastError();
}

hipError_t launch_gather_permuted(const aqe_record* aos, const PermSpec& perm, aqe_record* out, hipStream_t s) {
    if (perm.target == 0) return hipSuccess;
    hipLaunchKernelGGL(k_gather_permuted, dim3(grid_for(perm.target, kBlockThreads)), dim3(kBlockThreads), 0, s, aos, perm, out);
    return hipGetLastError();
}

hipError_t launch_update(QueryState* state, const double* vec, const FoldParams& p, int reset_state, hipStream_t s) {
    hipLaunchKernelGGL(k_update, dim3(1), dim3(64), 0, s, state, vec, p, reset_state);
    return hipGetLastError();
}

hipError_t launch_replay(const double* totals, uint32_t rounds, uint32_t has_topup, const FoldParams& fp,
                         const FinalizeParams& fin, QueryState* state, aqe_result* result, hipStream_t s) {
    hipLaunchKernelGGL(k_replay, dim3(1), dim3(64), 0, s, totals, rounds, has_topup, fp, fin, state, result);
    return hipGetLastError();
}

hipError_t launch_replay_batch(const ReplayItem* items, uint32_t n, const double* totals, uint64_t row_stride, hipStream_t s) {
    hipLaunchKernelGGL(k_replay_batch, dim3(n), dim3(64), 0, s, items, totals, static_cast<u64>(row_stride));
    return hipGetLastError();
}

hipError_t launch_finalize(const QueryState* state, const FinalizeParams& p, aqe_result* out, hipStream_t s) {
    hipLaunchKernelGGL(k_finalize, dim3(1), dim3(64), 0, s, state, p, out);
    return hipGetLastError();
}

hipError_t launch_gather(const aqe_record* aos, uint64_t shard_lo, const DevFamily* fams, uint32_t nfam,
                         uint64_t ntiles, aqe_record* out, int dense16, const uint32_t* perm, hipStream_t s) {
    if (ntiles == 0) return hipSuccess;
    unsigned grid = grid_for(ntiles, kWavesPerBlock);
    hipLaunchKernelGGL(k_gather, dim3(grid), dim3(kBlockThreads), 0, s, aos, static_cast<u64>(shard_lo), fams, nfam,
                       static_cast<u64>(ntiles), out, dense16, perm);
    return hipGetLastError();
}

hipError_t launch_gather_indexed(const aqe_record* aos, uint64_t shard_lo, const uint64_t* idx, uint64_t n,
                                 aqe_record* out, hipStream_t s) {
    if (n == 0) return hipSuccess;
    unsigned grid = grid_for(n, kBlockThreads);
    hipLaunchKernelGGL(k_gather_indexed, dim3(grid), dim3(kBlockThreads), 0, s, aos, static_cast<u64>(shard_lo), idx,
                       static_cast<u64>(n), out);
    return hipGetLastError();
}

hipError_t launch_id_bounds(const aqe_record* aos, uint64_t n, int64_t id_min, int64_t id_max, uint64_t* out, hipStream_t s) {
    hipLaunchKernelGGL(k_id_bounds, dim3(1), dim3(64), 0, s, aos, static_cast<u64>(n), static_cast<long long>(id_min),
                       static_cast<long long>(id_max), reinterpret_cast<u64*>(out));
    return hipGetLastError();
}

hipError_t launch_split_amount(const aqe_record* aos, double* amount, uint64_t n, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_split_amount, dim3(grid_for(n, kBlockThreads * 4)), dim3(kBlockThreads), 0, s, aos, amount,
                       static_cast<u64>(n));
    return hipGetLastError();
}

hipError_t launch_stride_view(const double* amount, uint64_t n, uint64_t shard_lo, uint64_t step, uint64_t M, uint64_t q0, double* out,
                              hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_stride_view, dim3(grid_for(n, kBlockThreads * 4)), dim3(kBlockThreads), 0, s, amount, static_cast<u64>(n),
                       static_cast<u64>(shard_lo), static_cast<u64>(step), static_cast<u64>(M), static_cast<u64>(q0), out);
    return hipGetLastError();
}

hipError_t launch_stride_view_keys(const int32_t* keys, uint64_t n, uint64_t shard_lo, uint64_t step, uint64_t M, uint64_t q0, int32_t* out,
                                   hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_stride_view_keys, dim3(grid_for(n, kBlockThreads * 4)), dim3(kBlockThreads), 0, s, keys, static_cast<u64>(n),
                       static_cast<u64>(shard_lo), static_cast<u64>(step), static_cast<u64>(M), static_cast<u64>(q0), out);
    return hipGetLastError();
}

hipError_t launch_synth(aqe_record* aos_or_null, double* amount, uint64_t n, uint64_t first_row, uint64_t seed,
                        hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_synth, dim3(grid_for(n, kBlockThreads * 4)), dim3(kBlockThreads), 0, s, aos_or_null, amount,
                       static_cast<u64>(n), static_cast<u64>(first_row), static_cast<u64>(seed));
    return hipGetLastError();
}

}  // namespace aqe
