// persist.hip — single-launch sweep of a multi-round (CLT) query with device-side early termination.
//
// The reference's CLT monitor (custom_bplus_db.cpp:885-1043) lets its fast/slow pointer threads poll an
// atomic<bool> should_stop on every iteration (DB.cpp:930, 987) while one of them recomputes the running
// statistics every check_interval samples.  On the GPU a launch costs 3-5 us and a grid-wide barrier 4-10 us,
// so a launch or a barrier per convergence step would cost more than the whole 10 M-row sweep (12 us).
// Instead ONE launch sweeps every round speculatively:
//
//   * tiles of all rounds form one list in round order; wave w owns tiles w, w+W, w+2W, ...  Waves never
//     wait for each other: while round r is being decided the chip is already sweeping rounds r+1, r+2...
//   * when the last wave of a workgroup leaves round r it publishes the workgroup's partial
//     (n, S-c n, Q) x {fast, slow} with write-through stores and draws a ticket (sharded counters);
//   * the wave that draws round r's last ticket is its DECIDER: it waits for round r-1's decision (already
//     under way, never the other way round, so no cycle), sums the partials in workgroup order
//     (bit-reproducible), folds them into the running Welford state, applies the CLT rules
//     (DB.cpp:936-961, 993-1016) and publishes continue/stop;
//   * every wave reads the stop word beside the loads of each tile (an sc1 load in the same vmcnt queue):
//     after a stop it sweeps nothing more and only hands in its remaining tickets, so every counter is
//     back at zero when the launch ends;
//   * the answer is the state after the FIRST round whose pooled moments satisfy the rule — work done
//     speculatively past that round is discarded, so results do not depend on timing.
//
// Hand-offs follow cdna_hip_programming.md Guideline 16 in its all-sc1 form: every shared word is written
// by ONE lane with 8-byte agent-scope stores, drained (s_waitcnt vmcnt(0)) before the ticket / decision
// that publishes it, and read with agent-scope loads.  Every spin is bounded.
#include "device_common.hpp"

namespace aqe {
namespace {

constexpr unsigned kCodeContinue = 1;
#define AQE_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Diagnostics: with a stamp buffer attached, every wave marks its own slots with plain stores (no
// contention), in 100 MHz s_memrealtime ticks.  Layout: [wave][8] then, per round r, [8] decider slots.
// wave slots: 0 start, 1 first tile swept, 2 last tile swept, 3 end; round slots: 3 decider chosen,
// 4 decider past its wait, 5 decider done, 2 (max over shards) shard reduce done.
__device__ __forceinline__ void stamp_wave(const PersistLaunch& P, unsigned slot, int lane) {
    if (P.stamps && lane == 0) {
        const u64 w = static_cast<u64>(blockIdx.x) * kWavesPerBlock + (threadIdx.x >> 6);
        P.stamps[w * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    }
}
__device__ __forceinline__ void stamp_max(const PersistLaunch& P, unsigned slot, int lane) {
    if (P.stamps && lane == 0) {
        const u64 W = static_cast<u64>(gridDim.x) * kWavesPerBlock;
        atomicMax(P.stamps + W * 8 + (slot - 8), __builtin_amdgcn_s_memrealtime());
    }
}
__device__ __forceinline__ void stamp_min(const PersistLaunch&, unsigned, int) {}

// does wave `w` (of W, a power of two) own a tile in [b0, b1)?  Its tiles are w, w+W, ...
__device__ __forceinline__ bool wave_has_tile(u64 w, u64 W, u64 b0, u64 b1) {
    const u64 first = b0 + ((w - b0) & (W - 1));  // smallest t >= b0 with t = w (mod W)
    return first < b1;
}

__device__ __forceinline__ bool block_has_tile(u64 b, u64 W, u64 b0, u64 b1) {
    return wave_has_tile(4 * b, W, b0, b1) || wave_has_tile(4 * b + 1, W, b0, b1) ||
           wave_has_tile(4 * b + 2, W, b0, b1) || wave_has_tile(4 * b + 3, W, b0, b1);
}

__device__ __forceinline__ void state_load(QueryState& st, const QueryState* g) {
    static_assert(sizeof(QueryState) % 8 == 0, "state is moved as 8-byte words");
    unsigned long long* d = reinterpret_cast<unsigned long long*>(&st);
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(g);
#pragma unroll
    for (unsigned i = 0; i < sizeof(QueryState) / 8; ++i) d[i] = __hip_atomic_load(s + i, AQE_RLX);
}

__device__ __forceinline__ void state_store(QueryState* g, const QueryState& st) {
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

constexpr int kMaxBlocksPerShard = 64;  // persist_grid <= 1024

// The decider of round r (one whole wave).  Deciders do not chain: each one announces that its round's
// shard partials are complete, waits until every earlier round has announced the same, and then replays
// the folds of rounds 0..r from the shard partials itself (the decision is a pure function of them).
// Exactly one decider finds that ITS round is the first to satisfy the stop rule (or is the last round):
// that one writes the state and the result and raises should_stop.  All others have nothing to publish.
__device__ void decide_round(const PersistLaunch& P, unsigned r, int lane, const uint16_t* lds_ex) {
    const unsigned long long tag = P.epoch << 8;
    stamp_max(P, 8 + 8 * r + 3, lane);
    if (lane == 0) __hip_atomic_store(&P.ctl->dec[r], tag | kCodeContinue, AQE_RLX);  // "round r is complete"
    // wait for rounds 0..r-1 (lane q watches round q); leave early if an earlier round already stopped
    bool abandoned = false;
    int timed_out = 0;
    for (unsigned spins = 0;; ++spins) {
        const bool mine = static_cast<unsigned>(lane) < r;
        const unsigned long long d = __hip_atomic_load(&P.ctl->dec[mine ? lane : r], AQE_RLX);
        const unsigned long long sw = __hip_atomic_load(&P.ctl->stop_word, AQE_RLX);
        if (sw == (tag | 1ull)) { abandoned = true; break; }
        if (__all(!mine || (d >> 8) == P.epoch)) break;
        if (spins > (1u << 22)) { timed_out = 1; break; }
        __builtin_amdgcn_s_sleep(1);
    }
    if (abandoned) return;
    stamp_max(P, 8 + 8 * r + 4, lane);
    // replay: lane L sums the 16 shard partials (ascending shard) of component c = L & 7 of round
    // q0 + (L >> 3); the folding then reads each round's seven totals through v_readlane.
    QueryState st{};
    bool stop = false;
    if (!timed_out) {
        const int c = lane & 7;
        for (unsigned q0 = 0; q0 <= r && !stop; q0 += 8) {
            const unsigned q = q0 + static_cast<unsigned>(lane >> 3);
            const bool act = c < 7 && q <= r;
            const uint16_t* ex = lds_ex + static_cast<size_t>(act ? q : 0) * (kPersistShards + 1);
            const double* sp = P.shard_partials + static_cast<size_t>(act ? q : 0) * kPersistShards * kVec;
            double x[kPersistShards];
#pragma unroll
            for (int sh = 0; sh < kPersistShards; ++sh) {
                const bool u = act && ex[sh] != 0;
                x[sh] = __hip_atomic_load(sp + (u ? sh * kVec + c : 0), AQE_RLX);
                if (!u) x[sh] = 0.0;
            }
            double tot = 0.0;
#pragma unroll
            for (int sh = 0; sh < kPersistShards; ++sh) tot += x[sh];
#pragma unroll 1
            for (int i = 0; i < 8 && q0 + i <= r; ++i) {  // wave-uniform trip count; source lanes live in SGPRs
                double vec[kVec];
#pragma unroll
                for (int cc = 0; cc < 7; ++cc) vec[cc] = read_lane_f64(tot, 8 * i + cc);
                vec[7] = 0.0;
                fold(st, vec, P.fold);
                stop = st.stop != 0;
                if (stop) {
                    if (q0 + i < r) return;  // an earlier round ends the query: its decider reports it
                    break;
                }
            }
        }
    } else {
        st.error = 1;
    }
    stop = stop || timed_out || r + 1 == P.rounds;
    if (!stop) return;
    if (lane == 0) {
        state_store(P.state, st);
        finalize(st, P.fin, P.result);  // the top-up launch, if it runs, rewrites the result
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&P.ctl->stop_word, tag | 1ull, AQE_RLX);
    }
    stamp_max(P, 8 + 8 * r + 5, lane);
}

// Last workgroup of shard `sh` to arrive in round r: sum the shard's workgroup partials (workgroups
// sh, sh+16, ... that own tiles of the round) in ascending workgroup order and publish ONE shard partial.
__device__ __forceinline__ void shard_reduce(const PersistLaunch& P, unsigned r, unsigned sh, int lane) {
    const u64 W = static_cast<u64>(gridDim.x) * kWavesPerBlock;
    const u64 b0 = P.round_begin[r], b1 = P.round_begin[r + 1];
    const unsigned per_shard = gridDim.x / kPersistShards;
    const double* part = P.partials + static_cast<size_t>(r) * gridDim.x * kVec;
    const int k = lane & 7, j = lane >> 3;
    double x[kMaxBlocksPerShard / 8];
    bool use[kMaxBlocksPerShard / 8];
#pragma unroll
    for (int i = 0; i < kMaxBlocksPerShard / 8; ++i) {  // member index m = j + 8 i, workgroup b = sh + 16 m
        const unsigned m = static_cast<unsigned>(j + 8 * i);
        const unsigned b = sh + kPersistShards * m;
        use[i] = k < 7 && m < per_shard && block_has_tile(b, W, b0, b1);
        x[i] = __hip_atomic_load(part + (use[i] ? static_cast<size_t>(b) * kVec + k : 0), AQE_RLX);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < kMaxBlocksPerShard / 8; ++i) s += use[i] ? x[i] : 0.0;
    s = class_sum8(s);
    if (lane < 7) __hip_atomic_store(P.shard_partials + (static_cast<size_t>(r) * kPersistShards + sh) * kVec + lane, s, AQE_RLX);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Last wave of this workgroup to leave round r: publish the workgroup's partial (unless the round was
// abandoned after a stop) and draw the workgroup's ticket; the shard's last workgroup reduces the shard
// and draws the shard's ticket; the last shard decides.
__device__ __forceinline__ void block_publish(const PersistLaunch& P, unsigned r, int lane, bool with_partial,
                                              double (*lds_part)[kWavesPerBlock][kVec], const uint16_t* lds_ex) {
    if (with_partial && lane < 7) {
        double s = lds_part[r][0][lane];
#pragma unroll
        for (int w = 1; w < kWavesPerBlock; ++w) s += lds_part[r][w][lane];
        __hip_atomic_store(P.partials + (static_cast<size_t>(r) * gridDim.x + blockIdx.x) * kVec + lane, s, AQE_RLX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint16_t* ex = lds_ex + static_cast<size_t>(r) * (kPersistShards + 1);
    const unsigned sh = blockIdx.x % kPersistShards;
    unsigned shard_last = 0;
    if (lane == 0) {
        unsigned* cs = &P.ctl->shard_cnt[r][sh][0];
        if (__hip_atomic_fetch_add(cs, 1u, AQE_RLX) + 1 == ex[sh]) {
            __hip_atomic_store(cs, 0u, AQE_RLX);
            shard_last = 1;
        }
    }
    if (!__builtin_amdgcn_readfirstlane(shard_last)) return;
    if (with_partial) shard_reduce(P, r, sh, lane);  // an abandoned round is never folded: tickets only
    stamp_max(P, 8 + 8 * r + 2, lane);
    unsigned decider = 0;
    if (lane == 0) {
        unsigned* ct = &P.ctl->top_cnt[r][0];
        if (__hip_atomic_fetch_add(ct, 1u, AQE_RLX) + 1 == ex[kPersistShards]) {
            __hip_atomic_store(ct, 0u, AQE_RLX);
            decider = 1;
        }
    }
    if (__builtin_amdgcn_readfirstlane(decider)) decide_round(P, r, lane, lds_ex);
}

// A wave leaves round r: hand its sums to the workgroup (LDS) and, if it is the workgroup's last wave in
// that round, publish.  with_partial=false after a stop: tickets only.
__device__ __forceinline__ void leave_round(const PersistLaunch& P, unsigned r, const Acc& acc, int lane, unsigned wave,
                                            bool with_partial, double (*lds_part)[kWavesPerBlock][kVec], unsigned* lds_cnt,
                                            const uint16_t* lds_ex) {
    if (with_partial) {
        const double v[7] = {static_cast<double>(acc.na), acc.sa, acc.qa, static_cast<double>(acc.nb), acc.sb, acc.qb,
                             static_cast<double>(acc.nv)};
        const double mine = wave_sum7(v, lane);  // lane 8c holds component c
        if ((lane & 7) == 0 && lane < 56) lds_part[r][wave][lane >> 3] = mine;
    }
    const u64 W = static_cast<u64>(gridDim.x) * kWavesPerBlock;
    const u64 b0 = P.round_begin[r], b1 = P.round_begin[r + 1];
    unsigned nw = 0;
#pragma unroll
    for (unsigned j = 0; j < kWavesPerBlock; ++j) nw += wave_has_tile(static_cast<u64>(blockIdx.x) * kWavesPerBlock + j, W, b0, b1) ? 1u : 0u;
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&lds_cnt[r], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old + 1 == nw) block_publish(P, r, lane, with_partial, lds_part, lds_ex);
}

__global__ __launch_bounds__(kBlockThreads, 4) void k_sweep_persist(PersistLaunch P) {
    __shared__ double lds_part[kMaxPersistRounds][kWavesPerBlock][kVec];
    __shared__ unsigned lds_cnt[kMaxPersistRounds];
    __shared__ DevFamily lds_fams[kMaxLdsFams];
    __shared__ uint16_t lds_ex[kMaxPersistRounds * (kPersistShards + 1)];
    for (unsigned i = threadIdx.x; i < P.rounds * (kPersistShards + 1); i += kBlockThreads) lds_ex[i] = P.expected[i];
    for (unsigned i = threadIdx.x; i < kMaxPersistRounds * kWavesPerBlock * kVec; i += kBlockThreads) (&lds_part[0][0][0])[i] = 0.0;
    if (threadIdx.x < kMaxPersistRounds) lds_cnt[threadIdx.x] = 0;
    const DevFamily* fams = stage_families(P.sw, lds_fams);
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const unsigned wave = threadIdx.x >> 6;
    const u64 W = static_cast<u64>(gridDim.x) * kWavesPerBlock;
    const u64 w = uniform64(static_cast<u64>(blockIdx.x) * kWavesPerBlock + wave);
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
            if (open && (!have || t >= P.round_begin[r + 1])) leave = true;  // round r is finished for this wave
            else if (!have) break;
            else while (t >= P.round_begin[r + 1]) ++r;                      // move to tile t's round
        } else {
            // A stop was published (necessarily for a round before r).  Hand in the tickets of round r and
            // of every later round this wave owns tiles in, sweeping nothing, so all counters return to zero.
            if (!open) {
                do { ++r; } while (r < P.rounds && !wave_has_tile(w, W, P.round_begin[r], P.round_begin[r + 1]));
                if (r >= P.rounds) break;
            }
            leave = true;
            with_partial = false;
        }
        if (leave) {
            leave_round(P, r, acc, lane, wave, with_partial, lds_part, lds_cnt, lds_ex);
            acc = Acc{};
            open = false;
            continue;
        }
        // should_stop (DB.cpp:930/987): one sc1 load issued beside the tile's own loads
        const unsigned long long sw = __hip_atomic_load(&P.ctl->stop_word, AQE_RLX);
        sweep_tile(P.sw, fams, t, lane, ~0ull, acc);
        if (t == w) stamp_wave(P, 1, lane);
        stamp_wave(P, 2, lane);
        open = true;
        t += W;
        if (sw == stop_tag) stopped = true;
    }
    stamp_wave(P, 3, lane);
}

}  // namespace

hipError_t launch_sweep_persist(const PersistLaunch& a, unsigned grid, hipStream_t s) {
    hipLaunchKernelGGL(k_sweep_persist, dim3(grid), dim3(kBlockThreads), 0, s, a);
    return hipGetLastError();
}

}  // namespace aqe
