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
//   * the wave that draws round r's last ticket is its DECIDER (decide_round below): one atomic OR settles which
//     decider judges which rounds, one batch of loads fetches the workgroup partials of the complete prefix,
//     and every round of the prefix is judged at once, lane q evaluating round q's stop rule
//     (DB.cpp:936-961, 993-1016).  Exactly one decider owns the first round that satisfies the rule (or the
//     last round): that one writes the state and the result and raises should_stop.  Deciders never wait for
//     each other, sums are taken in a fixed order (bit-reproducible), and a decision is a pure function of
//     the published partials, so speculative work past the stopping round cannot change the answer;
//   * every wave reads the stop word beside the loads of each tile (an sc1 load in the same vmcnt queue):
//     after a stop it sweeps nothing more and only hands in its remaining tickets, so every counter is back
//     at zero when the launch ends.
//
// Hand-offs follow cdna_hip_programming.md Guideline 16 in its all-sc1 form: every shared word is written by
// ONE lane (or one lane per word) with 8-byte agent-scope stores, drained (s_waitcnt vmcnt(0)) before the
// ticket / flag that publishes it, and read with agent-scope loads.  Every spin is bounded.
#include <hip/hip_ext.h>

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

#ifndef AQE_DECIDE_INLINE
#define AQE_DECIDE_INLINE __forceinline__
#endif
static_assert(kVec == 8, "a step of the flat partial list is 8 slots x 8 doubles = one 64-lane load");

// Decider scratch: the running per-lane sums of one batch of steps.  One decider per workgroup at a time.
__shared__ double lds_run[kDecSteps][64];
__shared__ unsigned lds_dec_lock;

// The decider of round r (one whole wave): the wave that drew the round's last ticket, so every workgroup
// partial of round r is published.
//
// Who judges which round is settled by ONE atomic: the decider ORs bit r into done_mask.  If every earlier
// round's bit was already set, this decider extends the complete prefix from r to the first still-open round
// p and is RESPONSIBLE for rounds [r, p); otherwise the decider of the lowest open round will extend the
// prefix over r when it arrives, and this one is finished.  Nobody waits for anybody.  The responsible decider
// loads the flat list of workgroup partials of rounds [0, p) in one batch (64 lanes x 8 B = one step of 8
// workgroups per load, coalesced), keeps a running sum per lane, and lane q picks the prefix total through
// round q out of LDS: all stop rules (DB.cpp:936-961, 993-1016) are evaluated at once, lane q judging round q.
// A decision is a pure function of the published partials, so two responsible deciders that overlap in time
// agree on the first stopping round and exactly one of them owns it.
//
// It runs once per round in the whole grid, so none of it may leak into the sweep: the descriptor pointer and
// the round are passed through an empty asm, which keeps hipcc from hoisting the decider's address arithmetic
// into every wave's prologue (and spilling it), and it reads the launch descriptor through the kernarg pointer
// only (scalar loads).  Inlined: as a called function it would save and restore callee-saved vector registers
// through scratch — a memory round trip on the way out, on the critical path of the launch.
__device__ AQE_DECIDE_INLINE void decide_round(KargPtr Kv, unsigned rv) {
    u64 kbits = uniform64(reinterpret_cast<u64>(Kv));
    unsigned r = __builtin_amdgcn_readfirstlane(rv);
    asm volatile("" : "+s"(kbits), "+s"(r));
    const KargPtr K = (KargPtr)kbits;
    const unsigned lane = threadIdx.x & 63;
    PersistCtl* const ctl = K->ctl;
    const unsigned rounds = K->rounds;
    const unsigned long long stop_tag = (K->epoch << 8) | 1ull;
    const bool totals_only = K->totals_only != 0;
    stamp_round(K->stamps, r, 3, lane);

    unsigned q_lo = r, q_hi = r + 1;  // rounds whose (prefix) totals this decider needs
    if (!totals_only) {
        const unsigned bit = 1u << r, full = rounds >= 32 ? ~0u : (1u << rounds) - 1u;
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_or(&ctl->done_mask, bit, AQE_RLX);
        const unsigned long long sw = __hip_atomic_load(&ctl->stop_word, AQE_RLX);
        old = __builtin_amdgcn_readfirstlane(old);
        const unsigned now = old | bit;
        // every round's decider comes here exactly once per launch: the last one leaves the mask at zero
        if (now == full && lane == 0) __hip_atomic_store(&ctl->done_mask, 0u, AQE_RLX);
        if (sw == stop_tag) return;                       // an earlier round already ended the query
        if ((old & (bit - 1u)) != bit - 1u) return;       // an earlier round is still open: its decider judges this one
        asm volatile("" ::: "memory");                    // the partial loads below stay behind the OR
        q_lo = 0;
        q_hi = now == ~0u ? 32u : static_cast<unsigned>(__builtin_ctz(~now));
    }
    const unsigned S0 = totals_only ? K->step_begin[r] : 0u, S1 = K->step_begin[q_hi];
    // lane q: the step that completes round q
    const unsigned my_last = K->step_begin[(lane < rounds ? lane : rounds - 1u) + 1u] - 1u;
    const bool judge = lane >= q_lo && lane < q_hi;

    int lock_failed = 0;
    if (lane == 0) {
        unsigned spins = 0;
        while (__hip_atomic_exchange(&lds_dec_lock, 1u, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0u) {
            if (++spins > (1u << 22)) { lock_failed = 1; break; }  // cannot happen: the holder never waits
            __builtin_amdgcn_s_sleep(2);
        }
    }
    lock_failed = __builtin_amdgcn_readfirstlane(lock_failed);

    // ---- the flat partial list, kDecSteps steps per batch of loads; lane = 8 j + c holds component c of the
    //      j-th workgroup of each step; `run` = sum of everything this lane has seen since S0 ----
    const double* const flat = K->partials;
    double run = 0.0;
    double tot[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll 1
    for (unsigned m0 = S0; m0 < S1; m0 += kDecSteps) {
        double x[kDecSteps];
#pragma unroll
        for (int m = 0; m < kDecSteps; ++m) x[m] = __hip_atomic_load(flat + static_cast<size_t>(m0 + m) * 64 + lane, AQE_RLX);
#pragma unroll
        for (int m = 0; m < kDecSteps; ++m) {
            if (m0 + m < S1) run += x[m];
            lds_run[m][lane] = run;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave writes and reads: LDS is in order
        const unsigned e = my_last - m0;                    // wraps to a large value before this batch
        if (judge && e < static_cast<unsigned>(kDecSteps)) {
#pragma unroll
            for (int cc = 0; cc < 7; ++cc) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) t += lds_run[e][8 * j + cc];  // workgroup classes in fixed order
                tot[cc] = t;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the next batch overwrites
    }
    if (lane == 0 && !lock_failed) __hip_atomic_store(&lds_dec_lock, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    stamp_round(K->stamps, r, 4, lane);

    if (totals_only) {  // multi-GPU form: hand the slot total out; the decision is taken after the all-reduce
        if (lane == r) {
            double* o = K->out_totals + static_cast<size_t>(r) * kVec;
#pragma unroll
            for (int cc = 0; cc < 7; ++cc) o[cc] = tot[cc];
            o[7] = 0.0;
        }
        return;
    }

    // ---- lane q holds the moments after round q: judge every round of the prefix at once ----
    int code = 0;
    FoldParams fp;
    fp.shift = K->fold.shift; fp.z = K->fold.z; fp.e = K->fold.e; fp.base = K->fold.base; fp.is_clt = K->fold.is_clt; fp.is_topup = 0; fp.pad = 0;
    if (fp.is_clt && judge) code = clt_rules(tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], fp);
    const unsigned long long stops = __ballot(code != 0);
    const unsigned first = stops ? static_cast<unsigned>(__builtin_ctzll(stops)) : ~0u;
    if (first < r) return;  // a round before this decider's range ends the query: the decider that owns it reports it
    unsigned last_round;
    if (first < q_hi) last_round = first;                 // the rule is satisfied after round `first`
    else if (q_hi == rounds) last_round = rounds - 1u;    // samples exhausted
    else return;                                          // the query goes on

    if (lane == last_round) {
        QueryState st{};
        st.n_a = tot[0]; st.sd_a = tot[1]; st.qd_a = tot[2];
        st.n_b = tot[3]; st.sd_b = tot[4]; st.qd_b = tot[5];
        st.n_p = tot[0] + tot[3]; st.sd_p = tot[1] + tot[4]; st.qd_p = tot[2] + tot[5];
        st.visited = tot[6];
        st.rounds = static_cast<int32_t>(last_round + 1u);
        st.converged = code;
        st.stop = code != 0;
        st.error = lock_failed;
        FinalizeParams fin;
        fin.n_global = K->fin.n_global; fin.pct = K->fin.pct; fin.shift = K->fin.shift; fin.agg = K->fin.agg;
        fin.convention = K->fin.convention; fin.is_exact = K->fin.is_exact; fin.is_clt = K->fin.is_clt;
        state_store(K->state, st);
        if (K->finalize_here) finalize(st, fin, K->result);  // else the top-up launch that follows writes the result
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // state and result are out before should_stop is
        __hip_atomic_store(&ctl->stop_word, stop_tag, AQE_RLX);
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
            const size_t slot = 8u * static_cast<size_t>(K->step_begin[r]) + ((blockIdx.x - K->part_first[r]) & (gridDim.x - 1u));
            __hip_atomic_store(P.partials + slot * kVec + lane, s, AQE_RLX);
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
    if (threadIdx.x == 0) lds_dec_lock = 0;
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

hipError_t launch_sweep_persist(const PersistLaunch& a, unsigned grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) hipExtLaunchKernelGGL(k_sweep_persist, dim3(grid), dim3(kPersistThreads), 0, s, ev0, ev1, 0, a);
    else hipLaunchKernelGGL(k_sweep_persist, dim3(grid), dim3(kPersistThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace aqe
