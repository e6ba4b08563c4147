// persist.hip — single-launch sweep of a multi-round (CLT) query with device-side early termination.
//
// The reference's CLT monitor (custom_bplus_db.cpp:885-1043) lets its fast/slow pointer threads poll an
// atomic<bool> should_stop on every iteration (DB.cpp:930, 987) while a monitoring thread recomputes the running
// statistics every check_interval samples.  On the GPU a launch costs 3-5 us and a grid-wide barrier 4-10 us,
// so a launch or a barrier per convergence step would cost more than the whole 10 M-row sweep (12 us).
// Instead ONE launch sweeps every round speculatively, and one wave plays the monitor:
//
//   * one workgroup of 16 waves per CU; every wave but one is a SWEEPER: tiles of all rounds form one list in
//     round order and sweeper v (of V) owns tiles v, v+V, v+2V, ...  Waves never wait for each other: while
//     round r is being judged the chip is already sweeping rounds r+1, r+2...
//   * when the last wave of a workgroup leaves round r it sums the workgroup's waves through LDS and publishes
//     the workgroup's partial (n, S-c n, Q) x {leader, others} into its slot of a flat list (round order), drains
//     the stores, then writes the slot's flag word (= the launch's epoch).  No counters, no atomics.
//   * the MONITOR is wave 0 of workgroup 0; it sweeps nothing.  It rehearses its fold once (instruction cache)
//     and then polls the flag words of the flat list from the first unjudged round on (4 loads per lane); the longest
//     prefix of complete rounds is found from the flags; if it grew, the steps of the new rounds are read (now
//     ordered after the flags) in one batch of coalesced loads (a "step" = 8 slots = 64 doubles = one wave
//     load) and folded into a running per-lane sum, lane q picks the prefix total through round q out of LDS,
//     and every new round is judged at once, lane q evaluating round q's stop rule (DB.cpp:936-961, 993-1016).
//     The first round that satisfies a rule — or the last round — ends the query: the monitor writes the state
//     and the result and raises should_stop.  Sums are taken in a fixed order (bit-reproducible), and a decision
//     is a pure function of the published partials, so speculative work past the stopping round cannot change
//     the answer;
//   * every wave reads the stop word beside the loads of each tile (an sc1 load in the same vmcnt queue) and
//     simply ends when it is raised.
//
// Hand-offs follow cdna_hip_programming.md Guideline 16 in its all-sc1 form: every shared word is written by
// ONE lane (or one lane per word) with 8-byte agent-scope stores, drained (s_waitcnt vmcnt(0)) before the
// flag that publishes it, and read with agent-scope loads issued after the flag was seen.  The only spin is
// the monitor's poll, and it is bounded.
#include <hip/hip_ext.h>

#include "device_common.hpp"

namespace aqe {
namespace {

#define AQE_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// The launch descriptor is indexed dynamically (round_begin[r], fams[i]).  Indexing the by-value kernel
// parameter would make hipcc copy the arrays to scratch; reading them through the kernarg segment pointer
// (constant address space) keeps them in scalar loads.
typedef const AQE_KARG PersistLaunch* KargPtr;
typedef const AQE_KARG DevFamily* KargFams;

static_assert(kVec == 8, "a step of the flat partial list is 8 slots x 8 doubles = one 64-lane load");
static_assert(kDecSteps == 32, "the monitor keeps one bit per step of its window in a 32-bit mask");

// Diagnostics: with a stamp buffer attached, every wave marks its own slots with plain stores (no
// contention), in 100 MHz s_memrealtime ticks.  Layout: [wave][8] then, per round r, [8] monitor slots.
// wave slots: 0 start, 1 first tile swept, 2 last tile swept, 3 end, 4 handed to the workgroup, 5 workgroup
// partial stored, 6 drained; round slots: 3 seen complete, 4 partials read, 7 rules evaluated, 5 judged (state and result stored), 6 (round 0) rehearsal.
__device__ __forceinline__ void stamp_wave(const PersistLaunch& P, unsigned slot, int lane) {
    if (P.stamps && lane == 0) {
        const u64 w = static_cast<u64>(blockIdx.x) * kPersistWaves + (threadIdx.x >> 6);
        P.stamps[w * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    }
}
__device__ __forceinline__ void stamp_round(unsigned long long* stamps, unsigned r, unsigned slot, unsigned lane) {
    if (stamps && lane == 0) {
        const u64 W = static_cast<u64>(gridDim.x) * kPersistWaves;
        stamps[W * 8 + 8 * r + slot] = __builtin_amdgcn_s_memrealtime();
    }
}

// does sweeper `v` (of V) own a tile in [b0, b1)?  Its tiles are v, v+V, ...; m0 = b0 mod V (from the host)
__device__ __forceinline__ bool sweeper_has_tile(unsigned v, unsigned V, u64 b0, unsigned m0, u64 b1) {
    const unsigned d = v >= m0 ? v - m0 : v + V - m0;  // smallest t >= b0 with t = v (mod V) is b0 + d
    return b0 + d < b1;
}

__device__ __forceinline__ void state_store(QueryState* g, const QueryState& st) {
    static_assert(sizeof(QueryState) % 8 == 0, "state is moved as 8-byte words");
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(&st);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(g);
#pragma unroll
    for (unsigned i = 0; i < sizeof(QueryState) / 8; ++i) __hip_atomic_store(d + i, s[i], AQE_RLX);
}

// Monitor scratch: the running per-lane sums of one window of steps (used by wave 0 of workgroup 0 only).
__shared__ double lds_run[kDecSteps][64];
__shared__ unsigned long long lds_t0;  // device clock at the monitor's start (PersistLaunch::want_ticks), parked here, not in a register

// The monitor's poll reads the flag words of a window of kDecSteps steps that starts at a round boundary S:
// lane L takes the flags of slots L, L + 64, L + 128, L + 192 of the window (4 loads, 8 registers).
__device__ __forceinline__ void flags_issue(const double* partials, unsigned S, unsigned lane, unsigned long long (&f)[4]) {
    const unsigned long long* fl = reinterpret_cast<const unsigned long long*>(partials) + (static_cast<size_t>(S) * 8 + lane) * kVec + 7;
#pragma unroll
    for (int k = 0; k < 4; ++k) f[k] = __hip_atomic_load(fl + static_cast<size_t>(k) * 64 * kVec, AQE_RLX);
}
// ... and counts the rounds from `judged` on whose every slot is published (a prefix, in round order; the
// window starts at round `judged`, which lies inside it).  sb_hi: lane q holds the end step of round q.
__device__ __forceinline__ unsigned flags_rounds(const unsigned long long (&f)[4], unsigned long long epoch, unsigned S, unsigned judged,
                                                 unsigned rounds, unsigned sb_hi, unsigned lane) {
    unsigned first_bad = 8u * kDecSteps;  // first unpublished slot of the window
#pragma unroll
    for (int k = 3; k >= 0; --k) {
        const unsigned long long m = __ballot(f[k] != epoch && f[k] != kSlotAlways);
        if (m) first_bad = 64u * static_cast<unsigned>(k) + static_cast<unsigned>(__builtin_ctzll(m));
    }
    const unsigned ready = S + first_bad / 8u;  // steps [S, ready) are published whole
    return judged + static_cast<unsigned>(__builtin_popcountll(__ballot(lane >= judged && lane < rounds && sb_hi <= ready)));
}

// One poll: the number of complete rounds (>= judged).
// kEpochArg: the launch's epoch comes as an argument instead of out of the descriptor (k_sweep_multi: the descriptors
// of a batch are written once; only the epoch changes from launch to launch).
template <bool kEpochArg = false>
__device__ __forceinline__ unsigned monitor_poll(KargPtr Kv, unsigned judged, unsigned long long epoch_arg = 0) {
    u64 kbits = uniform64(reinterpret_cast<u64>(Kv));
    asm volatile("" : "+s"(kbits));  // nothing of this is to be hoisted into every wave's prologue
    const KargPtr K = (KargPtr)kbits;
    const unsigned lane = threadIdx.x & 63;
    const unsigned rounds = K->rounds;
    const unsigned S = K->step_begin[judged];
    const unsigned sb_hi = K->step_begin[(lane < rounds ? lane : rounds - 1u) + 1u];
    unsigned long long f[4];
    flags_issue(K->partials, S, lane, f);
    return flags_rounds(f, kEpochArg ? epoch_arg : K->epoch, S, judged, rounds, sb_hi, lane);
}

struct FoldOut {
    double run;
    unsigned judged;
};

// The monitor folds rounds [p, p_new), which a poll has just seen complete, and judges them.
//
// The steps of those rounds are read now — after the flags were seen — in one batch of coalesced loads (lane
// 8 j + c takes component c of the j-th slot of each step) and folded into the running per-lane sum `run`
// (carried from fold to fold: the sum over every step since step 0); lane q picks the prefix total through
// round q out of LDS, and every new round is judged at once, lane q evaluating round q's stop rule
// (DB.cpp:936-961, 993-1016).  The next poll is in flight while the rounds are judged: if it shows more rounds
// complete, those are folded straight away.  Returns (run, rounds judged) if the query goes on.  If it ends
// here — a rule is satisfied, or the samples are exhausted — the monitor writes the state and the result,
// raises should_stop and ENDS ITS WAVE inside this function.
//
// Inlined into monitor_main: as a called function it would save and restore callee-saved registers through
// scratch — a memory round trip on the way out.
//
// warm != 0 (1: the estimate worked out beside the rules, 2: worked out after the decision — an early stop, the
// head form): a rehearsal while the monitor has nothing to do — same instructions, nothing folded, results
// written to a scratch area — so that the real call finds its code in the instruction cache.
template <bool kEpochArg = false>
__device__ __forceinline__ FoldOut monitor_fold(KargPtr Kv, unsigned pv, unsigned p_newv, double run, unsigned warm, unsigned long long epoch_arg = 0) {
    u64 kbits = uniform64(reinterpret_cast<u64>(Kv));
    asm volatile("" : "+s"(kbits));
    const KargPtr K = (KargPtr)kbits;
    unsigned p = __builtin_amdgcn_readfirstlane(pv), p_new = __builtin_amdgcn_readfirstlane(p_newv);
    const unsigned lane = threadIdx.x & 63;
    const unsigned rounds = K->rounds;
    const unsigned long long epoch = kEpochArg ? epoch_arg : K->epoch;
    const bool totals_only = K->totals_only != 0;
    const bool tslot = K->topup_slot != 0;              // the last slot is the top-up: summed on its own, never judged
    const unsigned rounds_j = rounds - (tslot ? 1u : 0u);
    const bool own_sum = totals_only ? lane < rounds : (tslot && lane == rounds - 1u);
    const unsigned ql = lane < rounds ? lane : rounds - 1u;
    const unsigned sb_lo = K->step_begin[ql], sb_hi = K->step_begin[ql + 1u];  // lane q: the steps of round q
    for (;;) {
        const unsigned S = K->step_begin[p], Sn = K->step_begin[p_new];
        const unsigned E = Sn - S;  // steps to fold, <= kDecSteps
        const double* const win = K->partials + static_cast<size_t>(S) * 64 + lane;
        stamp_round(K->stamps, warm ? 0u : p_new - 1u, warm ? 6 : 3, lane);

        double x[kDecSteps];
#pragma unroll
        for (int m = 0; m < kDecSteps; ++m) x[m] = __hip_atomic_load(win + static_cast<size_t>(m) * 64, AQE_RLX);
#pragma unroll
        for (int m = 0; m < kDecSteps; ++m) {
            if (static_cast<unsigned>(m) < E) {
                // totals form: every round is summed on its own
                if ((totals_only || tslot) && __ballot(own_sum && sb_hi > sb_lo && sb_lo == S + static_cast<unsigned>(m)) != 0) run = 0.0;
                run += x[m];
            }
            lds_run[m][lane] = run;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave writes and reads: LDS is in order
        // the next poll (the window behind the rounds being folded) is issued now that the partials are in: its
        // round trip overlaps the judging below.  (Issued beside the partial loads it would only see what the
        // poll before this fold saw, and fold the sweep's last microsecond round by round.)
        unsigned long long f_next[4];
        flags_issue(K->partials, Sn, lane, f_next);
        const bool fresh = warm ? lane == 0 : (lane >= p && lane < p_new);  // lane q: round q is one of the new ones
        double tot[7] = {0, 0, 0, 0, 0, 0, 0};
        if (fresh && (warm || sb_hi > sb_lo)) {
            const unsigned e = warm ? 0u : sb_hi - 1u - S;
#pragma unroll
            for (int cc = 0; cc < 7; ++cc) {
                double t = 0.0;
#pragma unroll
                for (int j = 0; j < 8; ++j) t += lds_run[e][8 * j + cc];  // workgroup classes in fixed order
                tot[cc] = t;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // read before the next fold overwrites
        if (warm) { tot[0] = 64.0; tot[1] = 8.0; tot[2] = 512.0; tot[3] = 64.0; tot[4] = 8.0; tot[5] = 512.0; tot[6] = 128.0; }
        stamp_round(K->stamps, warm ? 0u : p_new - 1u, warm ? 6 : 4, lane);

        if (totals_only) {  // multi-GPU form: hand the slot totals out; the decision is taken after the all-reduce
            if (fresh && !warm) {
                double* o = K->out_totals + static_cast<size_t>(lane) * kVec;
#pragma unroll
                for (int cc = 0; cc < 7; ++cc) o[cc] = tot[cc];
                o[7] = 0.0;
            }
            if (p_new == rounds && !warm) {
                stamp_round(K->stamps, p_new - 1u, 5, lane);
                __builtin_amdgcn_endpgm();
            }
        } else {
            // ---- lane q holds the moments after round q: judge every new round at once.  Each lane also works
            //      out the result its round would report: the two chains of f64 divisions and square roots
            //      are independent and overlap, instead of the estimate waiting for the decision ----
            int code = 0;
            FoldParams fp;
            fp.shift = K->fold.shift; fp.z = K->fold.z; fp.e = K->fold.e; fp.base = K->fold.base; fp.is_clt = K->fold.is_clt; fp.is_topup = 0; fp.pad = 0;
            FinalizeParams fin;
            fin.n_global = K->fin.n_global; fin.pct = K->fin.pct; fin.shift = K->fin.shift; fin.agg = K->fin.agg;
            fin.convention = K->fin.convention; fin.is_exact = K->fin.is_exact; fin.is_clt = K->fin.is_clt;
            const bool with_result = K->finalize_here != 0;
            QueryState st{};
            st.n_a = tot[0]; st.sd_a = tot[1]; st.qd_a = tot[2];
            st.n_b = tot[3]; st.sd_b = tot[4]; st.qd_b = tot[5];
            st.n_p = tot[0] + tot[3]; st.sd_p = tot[1] + tot[4]; st.qd_p = tot[2] + tot[5];
            st.visited = tot[6];
            aqe_result res{};
            // (only when this fold is certain to end the query; an early stop works its result out afterwards)
            const bool result_now = with_result && ((p_new == rounds && !tslot) || warm == 1u);  // (warm == 2 rehearses the late estimate below)
            double tup[7] = {0, 0, 0, 0, 0, 0, 0};  // the top-up slot's own total
            if (tslot) {
#pragma unroll
                for (int cc = 0; cc < 7; ++cc) tup[cc] = read_lane_f64(tot[cc], static_cast<int>(rounds - 1u));
            }
            if (fresh) {
                if (fp.is_clt && lane < rounds_j) code = clt_rules(tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], fp);
                if (result_now) res = make_result(st, fin);
            }
            stamp_round(K->stamps, warm ? 0u : p_new - 1u, 7, lane);
            const unsigned long long stops = __ballot(code != 0);
            unsigned last_round = ~0u;
            if (warm) last_round = 0u;
            else if (stops) last_round = static_cast<unsigned>(__builtin_ctzll(stops));  // the rule is satisfied after this round
            else if (p_new == rounds) last_round = rounds_j - 1u;                          // samples exhausted
            if (last_round != ~0u) {
                if (lane == last_round) {
                    st.rounds = static_cast<int32_t>(last_round + 1u);
                    st.converged = code;
                    st.stop = code != 0;
                    // DB.cpp:1032: too few rows collected -> the top-up is due.  Swept with the rounds (head form): it is
                    // added here.  Otherwise the launch that follows applies it and clears the mark; when the host did
                    // not enqueue one (it is rarely due), it sees the mark.
                    const bool goes_on = code == 0 && K->more_rounds;  // head form: out of rounds, not out of samples
                    bool due = K->topup_gate && st.n_p < static_cast<double>(fp.base / 4);
                    if (tslot && due && !goes_on) {  // the fold of a top-up vector (device_common.hpp, fold)
                        st.n_p += tup[0]; st.sd_p += tup[1]; st.qd_p += tup[2];
                        st.topup += tup[0];
                        st.visited += tup[6];
                        due = false;
                    }
                    if (with_result && !result_now) res = make_result(st, fin);
                    if (K->want_ticks && !warm) {
                        st.t0 = lds_t0;
                        res.kernel_ms = static_cast<double>(__builtin_amdgcn_s_memrealtime() - st.t0) * 1e-5;
                    }
                    res.rounds = st.rounds;
                    res.converged = code;
                    res.topup_pending = goes_on ? 2 : due ? 1 : 0;  // 2: the host launches the plan's remaining rounds
                    state_store(warm ? K->rehearsal_state : K->state, st);
                    if (with_result) *(warm ? K->rehearsal_result : K->result) = res;
                    if (!warm) {
                        // the host polls the pinned result instead of waiting for the end of the launch, several
                        // microseconds later: the check word tells it when every field has landed (kernels.hpp)
                        if (with_result) __hip_atomic_store(K->result_seq, result_check(res, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        if (code != 0) {  // waves are still sweeping: state and result are out before should_stop is
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            __hip_atomic_store(&K->ctl->stop_word, (epoch << 8) | 1ull, AQE_RLX);
                        }  // (samples exhausted: nobody is left to stop, and the end of the launch publishes the stores)
                    }
                }
                if (!warm) {
                    stamp_round(K->stamps, p_new - 1u, 5, lane);
                    __builtin_amdgcn_endpgm();
                }
            }
        }
        if (warm) return FoldOut{run, p};
        // the query goes on: what did the poll that rode along see?
        p = p_new;
        p_new = flags_rounds(f_next, epoch, Sn, p, rounds, sb_hi, lane);
        if (p_new == p) return FoldOut{run, p};
    }
}

// A wave leaves round r: hand its sums to the workgroup (LDS); the workgroup's last wave in that round sums
// the waves (wave order) and publishes the workgroup's slot: data, drain, flag.
__device__ __forceinline__ void leave_round(const PersistLaunch& P, KargPtr K, unsigned r, const Acc& acc, int lane, unsigned wave,
                                            double (*lds_part)[kPersistWaves][kVec], unsigned* lds_cnt) {
    const double v[7] = {static_cast<double>(acc.na), acc.sa, acc.qa, static_cast<double>(acc.nb), acc.sb, acc.qb,
                         static_cast<double>(acc.nv)};
    const double mine = wave_sum7(v, lane);  // lane 8c holds component c
    if ((lane & 7) == 0 && lane < 56) lds_part[r][wave][lane >> 3] = mine;
    const unsigned V = gridDim.x * kPersistWaves - 1u;           // sweepers: every wave but the monitor
    const unsigned v0 = blockIdx.x * kPersistWaves - 1u;         // sweeper id of this workgroup's wave 0 (wraps for the monitor)
    const u64 b0 = K->round_begin[r], b1 = K->round_begin[r + 1];
    const unsigned m0 = K->round_mod[r];
    unsigned nw = 0;
#pragma unroll
    for (unsigned j = 0; j < kPersistWaves; ++j) nw += (v0 + j != ~0u && sweeper_has_tile(v0 + j, V, b0, m0, b1)) ? 1u : 0u;
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&lds_cnt[r], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    stamp_wave(P, 4, lane);
    if (old + 1 != nw) return;
    // the i-th workgroup of the round's cyclic run owns slot 8 step_begin[r] + i
    const size_t slot = 8u * static_cast<size_t>(K->step_begin[r]) + ((blockIdx.x - K->part_first[r]) & (gridDim.x - 1u));
    double* const out = P.partials + slot * kVec;
    if (lane < 7) {
        double x[kPersistWaves];
#pragma unroll
        for (unsigned j = 0; j < kPersistWaves; ++j) x[j] = lds_part[r][j][lane];  // all reads in flight at once
        double s = 0.0;
#pragma unroll
        for (unsigned j = 0; j < kPersistWaves; ++j)  // only waves that swept tiles of the round wrote their slot
            s += (v0 + j != ~0u && sweeper_has_tile(v0 + j, V, b0, m0, b1)) ? x[j] : 0.0;
        __hip_atomic_store(out + lane, s, AQE_RLX);
    }
    stamp_wave(P, 5, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp_wave(P, 6, lane);
    if (lane == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + 7), static_cast<unsigned long long>(P.epoch), AQE_RLX);
}

constexpr unsigned long long kGiveUpTicks = 3000000000ull;  // 30 s of the 100 MHz device clock

// The monitor's whole life (wave 0 of workgroup 0; it sweeps nothing).
__device__ __forceinline__ void monitor_main(const PersistLaunch& P, KargPtr K) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_s_setprio(3);
    stamp_wave(P, 0, lane);
    if (lane == 0) lds_t0 = __builtin_amdgcn_s_memrealtime();  // (the query's device-clock timing, and the give-up bound below)
    unsigned judged = 0, polls = 0;  // rounds [0, judged) are folded and judged
    double run = 0.0;                // this lane's running sum over every step folded so far
    // nothing can be complete yet: rehearse the fold, so that the real one finds its code in the instruction cache
#pragma nounroll
    for (unsigned wm = 1; wm <= 2u; ++wm) (void)monitor_fold(K, 0u, 0u, 0.0, wm);
    for (;;) {
        unsigned complete = monitor_poll(K, judged);
        if (K->topup_slot != 0 && complete != K->rounds) complete = judged;  // that form is judged once, when everything is in
        if (complete > judged) {
            const FoldOut o = monitor_fold(K, judged, complete, run, 0u);  // does not return if the query ends here
            run = o.run;
            judged = o.judged;
        } else if ((++polls & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - lds_t0 > kGiveUpTicks) {
            // Cannot happen in a healthy launch: every workgroup publishes every round it owns tiles of.  The bound is
            // TIME (the device's 100 MHz clock, looked at every 1024 polls), not a poll count: this launch's other
            // workgroups may sit behind other streams' kernels for as long as those take.  Report it
            // (aqe_result.device_status) instead of hanging, and raise should_stop so the sweepers leave as well.
            if (lane == 0) {
                QueryState st{};
                st.error = 1;
                state_store(P.state, st);
                finalize(st, P.fin, P.result);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&P.ctl->stop_word, (P.epoch << 8) | 1ull, AQE_RLX);
            }
            break;
        }
    }
    stamp_wave(P, 3, lane);
}

template <bool kNT>
__global__ __launch_bounds__(kPersistThreads) void k_sweep_persist(PersistLaunch P) {
    __shared__ double lds_part[kMaxPersistRounds][kPersistWaves][kVec];
    __shared__ unsigned lds_cnt[kMaxPersistRounds];
    __shared__ DevFamily lds_fams[kMaxLdsFams];
    if (threadIdx.x < kMaxPersistRounds) lds_cnt[threadIdx.x] = 0;
    const KargPtr K = (KargPtr)__builtin_amdgcn_kernarg_segment_ptr();
    // family table: from the kernel arguments when it fits (scalar loads, nothing to wait for), else LDS
    const DevFamily* lfams = P.inline_fams ? nullptr : stage_families(P.sw, lds_fams);
    const KargFams kfams = K->fams;
    // The LDS tables above are first needed when a wave LEAVES a round, so with the family table in the
    // kernel arguments the barrier that publishes them is taken after the wave's first tile is in flight.
    // (Not in workgroup 0: the monitor must have its barrier behind it before it starts.)
    bool synced = !P.inline_fams || blockIdx.x == 0;
    if (synced) __syncthreads();

    const int lane = threadIdx.x & 63;
    const unsigned wave = threadIdx.x >> 6;
    const u64 w = uniform64(static_cast<u64>(blockIdx.x) * kPersistWaves + wave);
    if (w == 0) {
        monitor_main(P, K);
        return;
    }
    const u64 V = static_cast<u64>(gridDim.x) * kPersistWaves - 1u;  // sweepers
    const unsigned long long stop_tag = (P.epoch << 8) | 1ull;
    stamp_wave(P, 0, lane);

    // One loop, one place where a round is left (the publish code is large: a single call site keeps the
    // sweep's registers for the loads).  Sweeper v owns tiles v, v + V, v + 2 V, ...
    Acc acc;
    unsigned r = 0;
    u64 round_end = 0;  // end tile of round r
    bool open = false;  // the wave has swept at least one tile of round r and not yet left it
    u64 t = w - 1u;
    for (;;) {
        const bool have = t < P.ntiles;
        if (open && (!have || t >= round_end)) {  // round r is finished for this wave
            if (!synced) { __syncthreads(); synced = true; }
            leave_round(P, K, r, acc, lane, wave, lds_part, lds_cnt);
            acc = Acc{};
            open = false;
            continue;
        }
        if (!have) break;
        // should_stop (DB.cpp:930/987): one sc1 load issued beside the tile's own loads
        const unsigned long long sw = __hip_atomic_load(&P.ctl->stop_word, AQE_RLX);
        // Which family owns tile t, and which round is that?  Every dependent load here is a round trip on the
        // wave's critical path (the sweep of a 10 M-row table is a handful of them), so: the families' first tiles
        // are compared out of kernel-argument registers, and the family record itself names its round.
        if (lfams) {
            const DevFamily& F = lfams[find_family(lfams, P.sw.nfam, t)];
            r = __builtin_amdgcn_readfirstlane(F.round);
            round_end = uniform64(F.round_end);
            sweep_family<kNT>(P.sw, F, t, lane, ~0ull, acc);
        } else {
            const unsigned t32 = static_cast<unsigned>(t);
            unsigned i = 0;
#pragma unroll
            for (int k = 1; k < kPersistInlineFams; ++k) i += t32 >= P.fam_begin[k] ? 1u : 0u;
            const auto& F = kfams[i];
            r = F.round;
            round_end = F.round_end;
            sweep_family<kNT>(P.sw, F, t, lane, ~0ull, acc);
        }
        if (t + 1u == w) stamp_wave(P, 1, lane);
        stamp_wave(P, 2, lane);
        open = true;
        t += V;
        if (sw == stop_tag) break;  // the monitor ended the query: nothing is owed to anybody
    }
    if (!synced) __syncthreads();  // every wave of the workgroup takes the barrier exactly once
    stamp_wave(P, 3, lane);
}


// ---- k_sweep_multi: a BATCH of queries in ONE launch ------------------------------------------------------------
//
// A 10 M-row query is a 5 us sweep inside a launch whose fixed costs (start, hand-off, the monitor's decision tail)
// are twice that; the reference pays the analogous price per call by creating its worker threads per query
// (custom_bplus_db.cpp:918-1029).  Here a batch of Q independent queries shares one launch: the grid is cut into Q
// GROUPS of workgroups, group q runs query q exactly as a launch of its own would — its own descriptor (a
// PersistLaunch in device memory, written once when the batch is made), its own partial list, its own monitor wave
// (wave 0 of the group's first workgroup), its own should_stop word — and the start and the tails of all Q queries
// are paid once, side by side.  Only the launch epoch changes from launch to launch: it travels as an argument.
// A workgroup finds its place through wg_map[blockIdx.x] = query << 32 | group size << 16 | index in the group; the host
// keeps a group's workgroups contiguous and its blocks aligned, so that workgroup k of every group sits on compute die
// k mod 8 — where the same slice of other queries' tiles goes through the same L2 (plans.hip, build_multi).
__device__ __forceinline__ void leave_round_multi(KargPtr K, unsigned bid, unsigned G, unsigned long long epoch, unsigned r, const Acc& acc, int lane,
                                                  unsigned wave, double (*lds_part)[kPersistWaves][kVec], unsigned* lds_cnt) {
    const double v[7] = {static_cast<double>(acc.na), acc.sa, acc.qa, static_cast<double>(acc.nb), acc.sb, acc.qb,
                         static_cast<double>(acc.nv)};
    const double mine = wave_sum7(v, lane);  // lane 8c holds component c
    if ((lane & 7) == 0 && lane < 56) lds_part[r][wave][lane >> 3] = mine;
    const unsigned V = G * kPersistWaves - 1u;    // the group's sweepers: every wave but its monitor
    const unsigned v0 = bid * kPersistWaves - 1u;  // sweeper id of this workgroup's wave 0 (wraps for the monitor)
    const u64 b0 = K->round_begin[r], b1 = K->round_begin[r + 1];
    const unsigned m0 = K->round_mod[r];
    unsigned nw = 0;
#pragma unroll
    for (unsigned j = 0; j < kPersistWaves; ++j) nw += (v0 + j != ~0u && sweeper_has_tile(v0 + j, V, b0, m0, b1)) ? 1u : 0u;
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&lds_cnt[r], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old + 1 != nw) return;
    const size_t slot = 8u * static_cast<size_t>(K->step_begin[r]) + ((bid - K->part_first[r]) & (G - 1u));
    double* const out = K->partials + slot * kVec;
    if (lane < 7) {
        double x[kPersistWaves];
#pragma unroll
        for (unsigned j = 0; j < kPersistWaves; ++j) x[j] = lds_part[r][j][lane];
        double s = 0.0;
#pragma unroll
        for (unsigned j = 0; j < kPersistWaves; ++j) s += (v0 + j != ~0u && sweeper_has_tile(v0 + j, V, b0, m0, b1)) ? x[j] : 0.0;
        __hip_atomic_store(out + lane, s, AQE_RLX);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // data, drain, flag (Guideline 16)
    if (lane == 0) __hip_atomic_store(reinterpret_cast<unsigned long long*>(out + 7), epoch, AQE_RLX);
}

// The monitor of one group.  Its wait is bounded by the device clock, not by a poll count: groups of a large batch
// may sit behind other groups' workgroups for as long as those take.  On give-up it also raises should_stop, so the
// group's sweepers leave instead of finishing a query nobody will report.
constexpr unsigned long long kMultiGiveUpTicks = kGiveUpTicks;
__device__ __forceinline__ void monitor_main_multi(KargPtr K, unsigned long long epoch) {
    const int lane = threadIdx.x & 63;
    __builtin_amdgcn_s_setprio(3);
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
    if (K->want_ticks && lane == 0) lds_t0 = t_start;
    unsigned judged = 0;
    double run = 0.0;
    for (;;) {
        unsigned complete = monitor_poll<true>(K, judged, epoch);
        if (K->topup_slot != 0 && complete != K->rounds) complete = judged;  // that form is judged once, when everything is in
        if (complete > judged) {
            const FoldOut o = monitor_fold<true>(K, judged, complete, run, 0u, epoch);  // does not return if the query ends here
            run = o.run;
            judged = o.judged;
            continue;
        }
        __builtin_amdgcn_s_sleep(8);  // ~0.25 us: Q monitors poll side by side, and nothing here is on a critical path
        if (__builtin_amdgcn_s_memrealtime() - t_start > kMultiGiveUpTicks) {
            if (lane == 0) {  // report it (aqe_result.device_status) instead of hanging
                QueryState st{};
                st.error = 1;
                state_store(K->state, st);
                FinalizeParams fin;
                fin.n_global = K->fin.n_global; fin.pct = K->fin.pct; fin.shift = K->fin.shift; fin.agg = K->fin.agg;
                fin.convention = K->fin.convention; fin.is_exact = K->fin.is_exact; fin.is_clt = K->fin.is_clt;
                finalize(st, fin, K->result);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(&K->ctl->stop_word, (epoch << 8) | 1ull, AQE_RLX);
            }
            break;
        }
    }
}

template <bool kNT>
__global__ __launch_bounds__(kPersistThreads) void k_sweep_multi(const PersistLaunch* table, const unsigned long long* wg_map, unsigned long long epoch) {
    __shared__ double lds_part[kMaxPersistRounds][kPersistWaves][kVec];
    __shared__ unsigned lds_cnt[kMaxPersistRounds];
    __shared__ DevFamily lds_fams[kMaxLdsFams];
    if (threadIdx.x < kMaxPersistRounds) lds_cnt[threadIdx.x] = 0;
    const u64 me = uniform64(wg_map[blockIdx.x]);
    const unsigned G = static_cast<unsigned>(me >> 16) & 0xffffu, bid = static_cast<unsigned>(me) & 0xffffu;
    const KargPtr K = (KargPtr)(table + (me >> 32));
    SweepCommon sw;
    sw.amount = K->sw.amount; sw.shard_lo = K->sw.shard_lo; sw.fams = K->sw.fams; sw.nfam = K->sw.nfam; sw.has_where = K->sw.has_where;
    sw.wmin = K->sw.wmin; sw.wmax = K->sw.wmax; sw.shift = K->sw.shift; sw.dense16 = K->sw.dense16; sw.nt = kNT ? 1 : 0;
    const DevFamily* fams = stage_families(sw, lds_fams);
    if (sw.nfam > kMaxLdsFams) __syncthreads();  // (stage_families took the barrier otherwise: lds_cnt is published either way)

    const int lane = threadIdx.x & 63;
    const unsigned wave = threadIdx.x >> 6;
    const u64 w = uniform64(static_cast<u64>(bid) * kPersistWaves + wave);  // wave id within the group
    if (w == 0) {
        monitor_main_multi(K, epoch);
        return;
    }
    const u64 V = static_cast<u64>(G) * kPersistWaves - 1u;
    const u64 ntiles = K->ntiles;
    const unsigned nfam = sw.nfam;
    const unsigned long long stop_tag = (epoch << 8) | 1ull;
    const unsigned long long* const stop_word = &K->ctl->stop_word;

    Acc acc;
    unsigned r = 0;
    u64 round_end = 0;
    bool open = false;
    u64 t = w - 1u;
    for (;;) {
        const bool have = t < ntiles;
        if (open && (!have || t >= round_end)) {  // round r is finished for this wave
            leave_round_multi(K, bid, G, epoch, r, acc, lane, wave, lds_part, lds_cnt);
            acc = Acc{};
            open = false;
            continue;
        }
        if (!have) break;
        const unsigned long long sw_word = __hip_atomic_load(stop_word, AQE_RLX);  // should_stop (DB.cpp:930/987)
        const DevFamily& F = fams[find_family(fams, nfam, t)];
        r = __builtin_amdgcn_readfirstlane(F.round);
        round_end = uniform64(F.round_end);
        sweep_family<kNT>(sw, F, t, lane, ~0ull, acc);
        open = true;
        t += V;
        if (sw_word == stop_tag) break;  // this query's monitor ended it: nothing is owed to anybody
    }
}

}  // namespace

hipError_t launch_sweep_persist(const PersistLaunch& a, unsigned grid, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (a.sw.nt) {
        if (ev0) hipExtLaunchKernelGGL(k_sweep_persist<true>, dim3(grid), dim3(kPersistThreads), 0, s, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(k_sweep_persist<true>, dim3(grid), dim3(kPersistThreads), 0, s, a);
    } else {
        if (ev0) hipExtLaunchKernelGGL(k_sweep_persist<false>, dim3(grid), dim3(kPersistThreads), 0, s, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(k_sweep_persist<false>, dim3(grid), dim3(kPersistThreads), 0, s, a);
    }
    return hipGetLastError();
}

hipError_t launch_sweep_multi(const PersistLaunch* table, const unsigned long long* wg_map, unsigned long long epoch, unsigned grid, bool nt,
                              hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (nt) {
        if (ev0) hipExtLaunchKernelGGL(k_sweep_multi<true>, dim3(grid), dim3(kPersistThreads), 0, s, ev0, ev1, 0, table, wg_map, epoch);
        else hipLaunchKernelGGL(k_sweep_multi<true>, dim3(grid), dim3(kPersistThreads), 0, s, table, wg_map, epoch);
    } else {
        if (ev0) hipExtLaunchKernelGGL(k_sweep_multi<false>, dim3(grid), dim3(kPersistThreads), 0, s, ev0, ev1, 0, table, wg_map, epoch);
        else hipLaunchKernelGGL(k_sweep_multi<false>, dim3(grid), dim3(kPersistThreads), 0, s, table, wg_map, epoch);
    }
    return hipGetLastError();
}

}  // namespace aqe
