// persist.hip — single-launch sweep of a multi-round (CLT) query with device-side early termination.
//
// The reference's CLT monitor (custom_bplus_db.cpp:885-1043) lets its fast/slow pointer threads poll an
// atomic<bool> should_stop on every iteration (DB.cpp:930, 987) while one of them recomputes the running
// statistics every check_interval samples.  On the GPU a launch costs 3-5 us and a grid-wide barrier 4-10 us,
// so a launch or a barrier per convergence step would cost more than the whole 10 M-row sweep (12 us).
// Instead ONE launch sweeps every round speculatively:
//
//   * one workgroup of 16 waves per CU; tiles of all rounds form one list in round order and wave w owns
//     tiles w, w+W, w+2W, ... (W a power of two).  Waves never wait for each other: while round r is being
//     decided the chip is already sweeping rounds r+1, r+2...
//   * when the last wave of a workgroup leaves round r it sums the workgroup's waves through LDS, publishes
//     the workgroup's partial (n, S-c n, Q) x {fast, slow} with write-through stores and draws a ticket
//     (sharded counters: a same-address device atomic costs ~20 ns and serialises);
//   * the wave that draws round r's last ticket is its DECIDER.  In ONE batch of loads it fetches the round's
//     workgroup partials, the "complete" flags of the earlier rounds and their published round totals; it
//     sums in workgroup order (bit-reproducible), publishes its own round total + flag, and replays the folds
//     of rounds 0..r — every lane q <= r evaluating round q's stop rule (DB.cpp:936-961, 993-1016) on the
//     prefix sums in parallel.  Exactly one decider finds that ITS round is the first to satisfy the rule (or
//     is the last round): that one writes the state and the result and raises should_stop.  Deciders never
//     wait for each other in the common case, and a decision is a pure function of the published partials,
//     so speculative work past the stopping round cannot change the answer;
//   * every wave reads the stop word beside the loads of each tile (an sc1 load in the same vmcnt queue):
//     after a stop it sweeps nothing more and only hands in its remaining tickets, so every counter is back
//     at zero when the launch ends.
//
// Hand-offs follow cdna_hip_programming.md Guideline 16 in its all-sc1 form: every shared word is written by
// ONE lane (or one lane per word) with 8-byte agent-scope stores, drained (s_waitcnt vmcnt(0)) before the
// ticket / flag that publishes it, and read with agent-scope loads.  Every spin is bounded.
#include "device_common.hpp"

namespace aqe {
namespace {

#define AQE_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// The launch descriptor is indexed dynamically (round_begin[r], fams[i]).  Indexing the by-value kernel
// parameter would make hipcc copy the arrays to scratch; reading them through the kernarg segment pointer
// (constant address space) keeps them in scalar loads.
#define AQE_KARG __attribute__((address_space(4)))
typedef const AQE_KARG PersistLaunch* KargPtr;
typedef const AQE_KARG DevFamily* KargFams;

// Diagnostics: with a stamp buffer attached, every wave marks its own slots with plain stores (no
// contention), in 100 MHz s_memrealtime ticks.  Layout: [wave][8] then, per round r, [8] decider slots.
// wave slots: 0 start, 1 first tile swept, 2 last tile swept, 3 end; round slots: 3 decider chosen,
// 4 decider loads back, 5 decider done.
__device__ __forceinline__ void stamp_wave(const PersistLaunch& P, unsigned slot, int lane) {
    if (P.stamps && lane == 0) {
        const u64 w = static_cast<u64>(blockIdx.x) * kPersistWaves + (threadIdx.x >> 6);
        P.stamps[w * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    }
}
__device__ __forceinline__ void stamp_round(unsigned long long* stamps, unsigned r, unsigned slot, int lane) {
    if (stamps && lane == 0) {
        const u64 W = static_cast<u64>(gridDim.x) * kPersistWaves;
        stamps[W * 8 + 8 * r + slot] = __builtin_amdgcn_s_memrealtime();
    }
}

// does wave `w` (of W, a power of two) own a tile in [b0, b1)?  Its tiles are w, w+W, ...
__device__ __forceinline__ bool wave_has_tile(u64 w, u64 W, u64 b0, u64 b1) {
    const u64 first = b0 + ((w - b0) & (W - 1));  // smallest t >= b0 with t = w (mod W)
    return first < b1;
}

__device__ __forceinline__ void state_store(QueryState* g, const QueryState& st) {
    static_assert(sizeof(QueryState) % 8 == 0, "state is moved as 8-byte words");
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(&st);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(g);
#pragma unroll
    for (unsigned i = 0; i < sizeof(QueryState) / 8; ++i) __hip_atomic_store(d + i, s[i], AQE_RLX);
}

// xor-butterfly over the lanes that share (lane & 7): every lane ends with the sum of its 8-lane class.
__device__ __forceinline__ double class_sum8(double v) {
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}

constexpr int kDeciderLoads = kMaxPersistGrid / 8;  // workgroup partials per lane (8 lanes share a component)

// The decider of round r (one whole wave).
// Not inlined on purpose: it runs once per round in the whole grid, and inlining lets hipcc hoist its address
// arithmetic into every wave's prologue (and spill it).  It reads the launch descriptor through the kernarg
// pointer only.
__device__ __noinline__ void decide_round(KargPtr Kv, unsigned rv) {
    // arguments arrive in vector registers: make them provably wave-uniform so the descriptor is read with
    // scalar loads instead of a chain of dependent vector loads
    const KargPtr K = (KargPtr)uniform64(reinterpret_cast<u64>(Kv));
    const unsigned r = __builtin_amdgcn_readfirstlane(rv);
    const int lane = threadIdx.x & 63;
    PersistCtl* const ctl = K->ctl;
    const unsigned long long epoch = K->epoch;
    const unsigned rounds = K->rounds;
    double* const round_totals = K->round_totals;
    const unsigned long long tag = epoch << 8;
    const bool totals_only = K->totals_only != 0;
    stamp_round(K->stamps, r, 3, lane);
    if (!totals_only && __hip_atomic_load(&ctl->stop_word, AQE_RLX) == (tag | 1ull)) {  // an earlier round already ended the query
        if (lane == 0) __hip_atomic_store(&ctl->dec[r], tag | 1ull, AQE_RLX);
        return;
    }
    // the workgroups that own tiles of round r form one cyclic run [first, first + count) of workgroup ids
    const unsigned part_first = K->part_first[r], part_count = K->part_count[r], gmask = gridDim.x - 1;
    const int c = lane & 7, j = lane >> 3;
    const double* part = K->partials + static_cast<size_t>(r) * gridDim.x * kVec;

    // ---- one batch of loads: this round's workgroup partials, earlier rounds' flags and round totals ----
    double x[kDeciderLoads];
    bool use[kDeciderLoads];
#pragma unroll
    for (int m = 0; m < kDeciderLoads; ++m) {  // lane (c, j) takes workgroups j, j + 8, j + 16, ...
        const unsigned b = static_cast<unsigned>(j + 8 * m);
        use[m] = c < 7 && b < gridDim.x && ((b - part_first) & gmask) < part_count;
        x[m] = __hip_atomic_load(part + (use[m] ? static_cast<size_t>(b) * kVec + c : 0), AQE_RLX);
    }
    const bool watcher = !totals_only && static_cast<unsigned>(lane) < r;  // lane q < r watches round q
    unsigned long long flag = __hip_atomic_load(&ctl->dec[watcher ? lane : 0], AQE_RLX);
    double tot_q[7];  // lane q: the published totals of round q
#pragma unroll
    for (int cc = 0; cc < 7; ++cc)
        tot_q[cc] = __hip_atomic_load(round_totals + static_cast<size_t>(watcher ? lane : 0) * kVec + cc, AQE_RLX);

    // ---- this round's total: workgroups ascending within a lane, then the fixed butterfly over j ----
    double s = 0.0;
#pragma unroll
    for (int m = 0; m < kDeciderLoads; ++m) s += use[m] ? x[m] : 0.0;
    s = class_sum8(s);  // lanes with (lane & 7) == c hold component c
    stamp_round(K->stamps, r, 4, lane);
    if (totals_only) {  // multi-GPU form: hand the slot total out; the decision is taken after the all-reduce
        if (lane < kVec) K->out_totals[static_cast<size_t>(r) * kVec + lane] = lane < 7 ? s : 0.0;
        return;
    }

    // ---- publish the round total and the "complete" flag for later deciders (the last round has none) ----
    if (r + 1 < rounds) {
        if (lane < 7) __hip_atomic_store(round_totals + static_cast<size_t>(r) * kVec + lane, s, AQE_RLX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&ctl->dec[r], tag | 1ull, AQE_RLX);
    }

    // ---- earlier rounds must be complete; normally they are, otherwise wait (bounded) and re-read ----
    int timed_out = 0;
    if (!__all(!watcher || (flag >> 8) == epoch)) {
        for (unsigned spins = 0;; ++spins) {
            flag = __hip_atomic_load(&ctl->dec[watcher ? lane : 0], AQE_RLX);
            const unsigned long long sw = __hip_atomic_load(&ctl->stop_word, AQE_RLX);
            if (sw == (tag | 1ull)) return;
            if (__all(!watcher || (flag >> 8) == epoch)) break;
            if (spins > (1u << 22)) { timed_out = 1; break; }
            __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int cc = 0; cc < 7; ++cc)
            tot_q[cc] = __hip_atomic_load(round_totals + static_cast<size_t>(watcher ? lane : 0) * kVec + cc, AQE_RLX);
    }
    // lane r gets this round's totals
#pragma unroll
    for (int cc = 0; cc < 7; ++cc) {
        const double mine = read_lane_f64(s, cc);
        if (static_cast<unsigned>(lane) == r) tot_q[cc] = mine;
    }

    // ---- replay: prefix sums in round order (the order of the one-launch-per-round path); lane q keeps
    //      the state after round q and evaluates that round's stop rule; all rounds are judged at once ----
    double run[7] = {0, 0, 0, 0, 0, 0, 0}, mine[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (unsigned q = 0; q <= r; ++q) {
#pragma unroll
        for (int cc = 0; cc < 7; ++cc) {
            run[cc] += read_lane_f64(tot_q[cc], static_cast<int>(q));
            if (static_cast<unsigned>(lane) == q) mine[cc] = run[cc];
        }
    }
    int code = 0;
    FoldParams fp;
    fp.shift = K->fold.shift; fp.z = K->fold.z; fp.e = K->fold.e; fp.base = K->fold.base; fp.is_clt = K->fold.is_clt; fp.is_topup = 0; fp.pad = 0;
    if (fp.is_clt && static_cast<unsigned>(lane) <= r)
        code = clt_rules(mine[0], mine[1], mine[2], mine[3], mine[4], mine[5], fp);
    const unsigned long long stops = __ballot(code != 0);
    const unsigned first = stops ? static_cast<unsigned>(__builtin_ctzll(stops)) : ~0u;
    if (first < r) return;  // an earlier round ends the query: its own decider reports it
    if (!(first == r || r + 1 == rounds || timed_out)) return;  // the query goes on

    // ---- this round ends the query (rule satisfied, or samples exhausted) ----
    if (static_cast<unsigned>(lane) == r) {
        QueryState st{};
        st.n_a = mine[0]; st.sd_a = mine[1]; st.qd_a = mine[2];
        st.n_b = mine[3]; st.sd_b = mine[4]; st.qd_b = mine[5];
        st.n_p = mine[0] + mine[3]; st.sd_p = mine[1] + mine[4]; st.qd_p = mine[2] + mine[5];
        st.visited = mine[6];
        st.rounds = static_cast<int32_t>(r + 1);
        st.converged = code;
        st.stop = code != 0;
        st.error = timed_out;
        FinalizeParams fin;
        fin.n_global = K->fin.n_global; fin.pct = K->fin.pct; fin.shift = K->fin.shift; fin.agg = K->fin.agg;
        fin.convention = K->fin.convention; fin.is_exact = K->fin.is_exact; fin.is_clt = K->fin.is_clt;
        state_store(K->state, st);
        if (K->finalize_here) finalize(st, fin, K->result);  // else the top-up launch that follows writes the result  // the top-up launch, if it runs, rewrites the result
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // state and result are out before should_stop is
        __hip_atomic_store(&ctl->stop_word, tag | 1ull, AQE_RLX);
    }
    stamp_round(K->stamps, r, 5, lane);
}

// Last wave of this workgroup to leave round r: sum the workgroup's waves (wave order), publish the
// workgroup's partial (unless the round was abandoned after a stop) and draw the tickets.
__device__ __forceinline__ void block_publish(const PersistLaunch& P, KargPtr K, unsigned r, int lane, bool with_partial,
                                              double (*lds_part)[kPersistWaves][kVec], const uint16_t* lds_ex) {
    if (with_partial) {
        const u64 W = static_cast<u64>(gridDim.x) * kPersistWaves;
        const u64 b0 = K->round_begin[r], b1 = K->round_begin[r + 1];
        if (lane < 7) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < kPersistWaves; ++w)  // only waves that swept tiles of the round wrote their slot
                if (wave_has_tile(static_cast<u64>(blockIdx.x) * kPersistWaves + w, W, b0, b1)) s += lds_part[r][w][lane];
            __hip_atomic_store(P.partials + (static_cast<size_t>(r) * gridDim.x + blockIdx.x) * kVec + lane, s, AQE_RLX);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    const uint16_t* ex = lds_ex + static_cast<size_t>(r) * (kPersistShards + 1);
    const unsigned sh = blockIdx.x % kPersistShards;
    unsigned decider = 0;
    if (lane == 0) {
        unsigned* cs = &P.ctl->shard_cnt[r][sh][0];
        if (__hip_atomic_fetch_add(cs, 1u, AQE_RLX) + 1 == ex[sh]) {
            __hip_atomic_store(cs, 0u, AQE_RLX);
            unsigned* ct = &P.ctl->top_cnt[r][0];
            if (__hip_atomic_fetch_add(ct, 1u, AQE_RLX) + 1 == ex[kPersistShards]) {
                __hip_atomic_store(ct, 0u, AQE_RLX);
                decider = 1;
            }
        }
    }
    if (__builtin_amdgcn_readfirstlane(decider)) decide_round(K, r);
}

// A wave leaves round r: hand its sums to the workgroup (LDS) and, if it is the workgroup's last wave in
// that round, publish.  with_partial=false after a stop: tickets only.
__device__ __forceinline__ void leave_round(const PersistLaunch& P, KargPtr K, unsigned r, const Acc& acc, int lane, unsigned wave,
                                            bool with_partial, double (*lds_part)[kPersistWaves][kVec], unsigned* lds_cnt,
                                            const uint16_t* lds_ex) {
    if (with_partial) {
        const double v[7] = {static_cast<double>(acc.na), acc.sa, acc.qa, static_cast<double>(acc.nb), acc.sb, acc.qb,
                             static_cast<double>(acc.nv)};
        const double mine = wave_sum7(v, lane);  // lane 8c holds component c
        if ((lane & 7) == 0 && lane < 56) lds_part[r][wave][lane >> 3] = mine;
    }
    const u64 W = static_cast<u64>(gridDim.x) * kPersistWaves;
    const u64 b0 = K->round_begin[r], b1 = K->round_begin[r + 1];
    unsigned nw = 0;
#pragma unroll
    for (unsigned j = 0; j < kPersistWaves; ++j) nw += wave_has_tile(static_cast<u64>(blockIdx.x) * kPersistWaves + j, W, b0, b1) ? 1u : 0u;
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&lds_cnt[r], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old + 1 == nw) block_publish(P, K, r, lane, with_partial, lds_part, lds_ex);
}

__global__ __launch_bounds__(kPersistThreads) void k_sweep_persist(PersistLaunch P) {
    __shared__ double lds_part[kMaxPersistRounds][kPersistWaves][kVec];
    __shared__ unsigned lds_cnt[kMaxPersistRounds];
    __shared__ uint16_t lds_ex[kMaxPersistRounds * (kPersistShards + 1)];
    __shared__ DevFamily lds_fams[kMaxLdsFams];
    for (unsigned i = threadIdx.x; i < P.rounds * (kPersistShards + 1); i += kPersistThreads) lds_ex[i] = P.expected[i];
    if (threadIdx.x < kMaxPersistRounds) lds_cnt[threadIdx.x] = 0;
    const KargPtr K = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    if (P.totals_only && blockIdx.x == 0 && threadIdx.x < P.rounds * kVec) {  // slots with no tile on this shard
        const unsigned r0 = threadIdx.x / kVec;
        if (K->round_begin[r0 + 1] == K->round_begin[r0]) P.out_totals[threadIdx.x] = 0.0;
    }
    // family table: from the kernel arguments when it fits (scalar loads, nothing to wait for), else LDS
    const DevFamily* lfams = P.inline_fams ? nullptr : stage_families(P.sw, lds_fams);
    const KargFams kfams = K->fams;
    // The LDS tables above are first needed when a wave LEAVES a round, so with the family table in the
    // kernel arguments the barrier that publishes them is taken after the wave's first tile is in flight.
    bool synced = !P.inline_fams;
    if (synced) __syncthreads();

    const int lane = threadIdx.x & 63;
    const unsigned wave = threadIdx.x >> 6;
    const u64 W = static_cast<u64>(gridDim.x) * kPersistWaves;
    const u64 w = uniform64(static_cast<u64>(blockIdx.x) * kPersistWaves + wave);
    const unsigned long long stop_tag = (P.epoch << 8) | 1ull;
    stamp_wave(P, 0, lane);

    // One loop, one place where a round is left (the publish/decide code is large: a single call site
    // keeps the sweep's registers for the loads).
    Acc acc;
    unsigned r = 0;
    bool open = false;     // the wave has swept at least one tile of round r and not yet left it
    bool stopped = false;  // a stop was observed: only tickets from here on
    u64 t = w;
    for (;;) {
        bool leave = false, with_partial = true;
        if (!stopped) {
            const bool have = t < P.ntiles;
            if (open && (!have || t >= K->round_begin[r + 1])) leave = true;  // round r is finished for this wave
            else if (!have) break;
            else while (t >= K->round_begin[r + 1]) ++r;                     // move to tile t's round
        } else {
            // A stop was published (necessarily for a round before r).  Hand in the tickets of round r and
            // of every later round this wave owns tiles in, sweeping nothing, so all counters return to zero.
            if (!open) {
                do { ++r; } while (r < P.rounds && !wave_has_tile(w, W, K->round_begin[r], K->round_begin[r + 1]));
                if (r >= P.rounds) break;
            }
            leave = true;
            with_partial = false;
        }
        if (leave) {
            if (!synced) { __syncthreads(); synced = true; }
            leave_round(P, K, r, acc, lane, wave, with_partial, lds_part, lds_cnt, lds_ex);
            acc = Acc{};
            open = false;
            continue;
        }
        // should_stop (DB.cpp:930/987): one sc1 load issued beside the tile's own loads
        const unsigned long long sw = __hip_atomic_load(&P.ctl->stop_word, AQE_RLX);
        if (lfams) sweep_tile(P.sw, lfams, t, lane, ~0ull, acc); else sweep_tile(P.sw, kfams, t, lane, ~0ull, acc);
        if (t == w) stamp_wave(P, 1, lane);
        stamp_wave(P, 2, lane);
        open = true;
        t += W;
        if (sw == stop_tag) stopped = true;
    }
    if (!synced) __syncthreads();  // every wave of the workgroup takes the barrier exactly once
    stamp_wave(P, 3, lane);
}

}  // namespace

hipError_t launch_sweep_persist(const PersistLaunch& a, unsigned grid, hipStream_t s) {
    hipLaunchKernelGGL(k_sweep_persist, dim3(grid), dim3(kPersistThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace aqe
