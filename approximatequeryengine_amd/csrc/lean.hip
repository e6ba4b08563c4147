// lean.hip — k_sweep_lean / k_sweep_lean_multi: every round of a multi-round (CLT) query in ONE lean launch, judged once
// at the end — and a batch of such queries in one launch, a group of workgroups each.
//
// The reference's monitor (custom_bplus_db.cpp:885-1043) re-evaluates its rules while the pointer threads are still
// walking, so that they can stop early (DB.cpp:930, 987).  k_sweep_persist (persist.hip) keeps that shape: a monitor
// wave judges rounds as they complete and raises should_stop.  On a table of 10 M rows there is nothing left to stop:
// every wave holds its tile in flight from the first microsecond, the sweep is over after five, and what the query then
// waits for is the hand-off.  This kernel is for those queries — and for longer sweeps that are predicted not to stop
// early (plans.hip, launch_form) — provided every family of the plan is a plain run of rows, which is what strided
// pointers read through their stride-major views are.  It is built for a short critical path:
//
//   * no monitor wave, no polling, no stop word: every wave sweeps.  Workgroup b owns the tiles [b K, (b + 1) K) of the
//     launch's list (K = tiles / workgroups, rounded up; its wave j takes the j-th, (j + 16)-th, ... of them: the 16 waves
//     stream one window of the column), so it sweeps tiles of one round, or of a few consecutive ones; it publishes ONE
//     56-byte partial per round it swept tiles of (sc1 stores, drained) and draws ONE ticket (sharded counters, as k_round);
//   * the workgroup that draws the last ticket sums every round's slots of the flat partial list — fewer than
//     workgroups + rounds of them, whatever the size of the sweep — in a fixed order (bit-reproducible), 64 threads per
//     round, and wave 0 then judges EVERY round at once, lane q evaluating the rules on the moments through round q
//     (DB.cpp:936-961, 993-1016).  The first round that satisfies a rule — or the last — is the query's answer: a
//     decision is a pure function of the partials, so the rounds swept beyond it change nothing;
//   * the tile decode is a dozen instructions: lane i of every wave holds runs i and i + 64 of the plan in registers (one batch of
//     loads in the prologue), the run of a tile is found by one wave-wide comparison and read with v_readlane; tiles
//     that lie inside their run — all but a run's last — take no masks.  With four waves per SIMD every instruction of
//     a wave costs ~8 ns of wall time while all waves do the same thing, which they do at the start and at the end of a
//     short query: k_sweep_persist's general tile path (~500 vector instructions of window arithmetic per tile) had the
//     first tile of a 10 M-row query folded after 3.3 us and the last after 8.3, where a lean loop needs 0.4 and 3.7
//     (tools/exp_latency.hip) — 17.7 us per launch against 9.5.
//
// Plans of more runs than the lanes hold (up to 512: 64 ... 256 pointers on 10 M rows) take the WIDE instantiation: the
// table goes through LDS, one copy per workgroup, and a tile's run is found by bisection.
//
// Same forms as the persistent sweep (PersistLaunch): decisions in the kernel, the head form (rounds + the top-up as
// one more slot, `more_rounds`), totals only (multi-GPU: every slot's total for the all-reduce).  Nothing in it waits
// for another workgroup, so the groups of a batch (k_sweep_lean_multi) may outnumber the compute units.
#include <hip/hip_ext.h>

#include "device_common.hpp"

namespace aqe {
namespace {

#define AQE_RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

typedef const AQE_KARG LeanLaunch* LeanKarg;

// Diagnostics (builds with -DAQE_LEAN_STAMPS only; tools/stamp_lean.py): s_memrealtime marks, 100 MHz.
// [wave][8]: 0 entry, 1 table in registers, 2 first tile folded, 3 sweep done, 4 sums handed to the workgroup, 5 partial out
// (wave 0), 6 ticket drawn (wave 0);  then [8] of the folding workgroup: 1 rounds summed, 2 judged
#ifdef AQE_LEAN_STAMPS
__device__ unsigned long long g_lean_stamps[(kMaxPersistGrid * kPersistWaves + 1) * 8];
#define LEAN_STAMP(slot) do { if (lane == 0) g_lean_stamps[(static_cast<size_t>(blockIdx.x) * kPersistWaves + wave) * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define LEAN_STAMP_FOLD(slot) do { if (threadIdx.x == 0) g_lean_stamps[static_cast<size_t>(kMaxPersistGrid) * kPersistWaves * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LEAN_STAMP(slot) do { } while (0)
#define LEAN_STAMP_FOLD(slot) do { } while (0)
#endif

__device__ __forceinline__ void lean_state_store(QueryState* g, const QueryState& st) {
    static_assert(sizeof(QueryState) % 8 == 0, "state is moved as 8-byte words");
    const unsigned long long* s = reinterpret_cast<const unsigned long long*>(&st);
    unsigned long long* d = reinterpret_cast<unsigned long long*>(g);
#pragma unroll
    for (unsigned i = 0; i < sizeof(QueryState) / 8; ++i) __hip_atomic_store(d + i, s[i], AQE_RLX);
}

__device__ __forceinline__ u64 read_lane_u64(u64 v, unsigned src_lane) {
    const unsigned lo = __builtin_amdgcn_readlane(static_cast<unsigned>(v), src_lane);
    const unsigned hi = __builtin_amdgcn_readlane(static_cast<unsigned>(v >> 32), src_lane);
    return (static_cast<u64>(hi) << 32) | lo;
}

__device__ __forceinline__ u64* lean_t0_word(unsigned* counter) { return reinterpret_cast<u64*>(counter + kShards * kShardStride + 8); }

struct __attribute__((packed, aligned(8))) Row2 { double x, y; };

// One tile = up to 1024 consecutive rows from `base` (two rows per lane per 16-byte load, eight loads in flight);
// `rem` rows of the run (or of the block) are left from `base` on.
template <bool kNT>
__device__ __forceinline__ void lean_tile(const double* base, unsigned rem, const double* safe, int lane, int has_where, double wmin, double wmax,
                                          double shift, TileAcc& ta) {
    Row2 v2[kTileUnroll];
    if (rem >= static_cast<unsigned>(kDenseTileOrdinals)) {  // inside the run: no masks
        const Row2* const p = reinterpret_cast<const Row2*>(base) + lane;
#pragma unroll
        for (int k = 0; k < kTileUnroll; ++k) {
            if (kNT) {  // a sweep beyond the Infinity Cache: past the caches (device_common.hpp, sweep_family)
                v2[k].x = __builtin_nontemporal_load(&p[k * 64].x);
                v2[k].y = __builtin_nontemporal_load(&p[k * 64].y);
            } else {
                v2[k] = p[k * 64];
            }
        }
        ta.nv = 2u * kTileUnroll;
        if (has_where) {
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {
                const double x = v2[k].x, y = v2[k].y;
                const bool px = x >= wmin && x <= wmax, py = y >= wmin && y <= wmax;  // inclusive both ends, DB.cpp:329
                const double dx = px ? x - shift : 0.0, dy = py ? y - shift : 0.0;
                ta.n += (px ? 1u : 0u) + (py ? 1u : 0u);
                ta.s += dx; ta.q += dx * dx;
                ta.s += dy; ta.q += dy * dy;
            }
        } else {
            ta.n = 2u * kTileUnroll;
#pragma unroll
            for (int k = 0; k < kTileUnroll; ++k) {
                const double dx = v2[k].x - shift, dy = v2[k].y - shift;
                ta.s += dx; ta.q += dx * dx;
                ta.s += dy; ta.q += dy * dy;
            }
        }
        return;
    }
    // the last tile of a run — every tile of a run of 1000-row blocks: the same 16-byte loads, a pair masked where it
    // leaves the run and never addressed there (such a lane reads rows 0 and 1 of the column instead)
    bool ok[kTileUnroll];
#pragma unroll
    for (int k = 0; k < kTileUnroll; ++k) {
        const unsigned oi = 2u * static_cast<unsigned>(lane) + 128u * static_cast<unsigned>(k);
        ok[k] = oi + 1u < rem;  // both rows of the pair inside the run
        const Row2* const p = reinterpret_cast<const Row2*>(ok[k] ? base + oi : safe);
        if (kNT) {
            v2[k].x = __builtin_nontemporal_load(&p->x);
            v2[k].y = __builtin_nontemporal_load(&p->y);
        } else {
            v2[k] = *p;
        }
    }
#pragma unroll
    for (int k = 0; k < kTileUnroll; ++k) {
        const double x = v2[k].x, y = v2[k].y;
        const bool px = ok[k] && (!has_where || (x >= wmin && x <= wmax)), py = ok[k] && (!has_where || (y >= wmin && y <= wmax));
        const double dx = px ? x - shift : 0.0, dy = py ? y - shift : 0.0;
        ta.nv += ok[k] ? 2u : 0u;
        ta.n += (px ? 1u : 0u) + (py ? 1u : 0u);
        ta.s += dx; ta.q += dx * dx;
        ta.s += dy; ta.q += dy * dy;
    }
    if (rem & 1u) {  // (wave-uniform) a run of odd length ends in half a pair: that row on its own, folded by lane 0
        const double x = base[rem - 1u];
        const bool px = lane == 0 && (!has_where || (x >= wmin && x <= wmax));
        const double dx = px ? x - shift : 0.0;
        ta.nv += lane == 0 ? 1u : 0u;
        ta.n += px ? 1u : 0u;
        ta.s += dx; ta.q += dx * dx;
    }
}

// Wave 0 of the folding workgroup, lane q holding the moments through round q (or a slot's own total where the form
// asks for that): the decision and the result.  The rules and what follows them are the monitor's (persist.hip,
// monitor_fold), evaluated once, for every round at the same time.  T: the launch's tail, in LDS.
__device__ __forceinline__ void lean_judge(const LeanTail& T, const double (&tot)[7], unsigned lane, unsigned long long t0, unsigned long long epoch, unsigned long long* res_words) {
    const unsigned rounds = T.rounds;
    const bool tslot = T.topup_slot != 0;  // the last slot is the top-up: summed on its own, never judged
    const unsigned rounds_j = rounds - (tslot ? 1u : 0u);
    if (T.totals_only) {  // multi-GPU form: hand the slot totals out; the decision is taken after the all-reduce
        if (lane < rounds) {
            double* o = T.out_totals + static_cast<size_t>(lane) * kVec;
#pragma unroll
            for (int cc = 0; cc < 7; ++cc) o[cc] = tot[cc];
            o[7] = 0.0;
        }
        return;
    }
    const FoldParams fp = T.fold;
    const FinalizeParams fin = T.fin;
    const bool with_result = T.finalize_here != 0;
    QueryState st{};
    st.n_a = tot[0]; st.sd_a = tot[1]; st.qd_a = tot[2];
    st.n_b = tot[3]; st.sd_b = tot[4]; st.qd_b = tot[5];
    st.n_p = tot[0] + tot[3]; st.sd_p = tot[1] + tot[4]; st.qd_p = tot[2] + tot[5];
    st.visited = tot[6];
    double tup[7] = {0, 0, 0, 0, 0, 0, 0};  // the top-up slot's own total
    if (tslot) {
#pragma unroll
        for (int cc = 0; cc < 7; ++cc) tup[cc] = read_lane_f64(tot[cc], static_cast<int>(rounds - 1u));
    }
    int code = 0;
    aqe_result res{};
    // every lane works its round's estimate out beside the rules: the two chains of f64 operations overlap (with a
    // top-up slot the estimate depends on the decision and follows it)
    const bool result_now = with_result && !tslot;
#ifndef AQE_ABL_NOJUDGE
    if (lane < rounds_j) {
        if (fp.is_clt) code = clt_rules(tot[0], tot[1], tot[2], tot[3], tot[4], tot[5], fp);
        if (result_now) res = make_result(st, fin);
    }
#else
    res.sum = st.sd_p; res.n = static_cast<uint64_t>(st.n_p);
#endif
    const unsigned long long stops = __ballot(code != 0);
    const unsigned last_round = stops ? static_cast<unsigned>(__builtin_ctzll(stops)) : rounds_j - 1u;  // the rule holds after this round / samples exhausted
    if (lane != last_round) return;
    st.rounds = static_cast<int32_t>(last_round + 1u);
    st.converged = code;
    st.stop = code != 0;
    // DB.cpp:1032: too few rows collected -> the top-up is due.  Swept with the rounds (head form): it is added here.
    // Otherwise the result is marked and the host launches it (rarely due).
    const bool goes_on = code == 0 && T.more_rounds;  // head form: out of rounds, not out of samples
    bool due = T.topup_gate && st.n_p < static_cast<double>(fp.base / 4);
    if (tslot && due && !goes_on) {  // the fold of a top-up vector (device_common.hpp, fold)
        st.n_p += tup[0]; st.sd_p += tup[1]; st.qd_p += tup[2];
        st.topup += tup[0];
        st.visited += tup[6];
        due = false;
    }
    if (with_result && !result_now) res = make_result(st, fin);
    if (T.want_ticks) {
        st.t0 = t0;
        res.kernel_ms = static_cast<double>(__builtin_amdgcn_s_memrealtime() - st.t0) * 1e-5;
    }
    res.rounds = st.rounds;
    res.converged = code;
    res.topup_pending = goes_on ? 2 : due ? 1 : 0;  // 2: the host launches the plan's remaining rounds
#ifdef AQE_ABL_NOSTORE  // (ablation: everything computed, ONE 8-byte store to device memory)
    T.state->n_a = res.value + res.ci_lower + st.n_p + static_cast<double>(result_check(res, epoch) & 0xffu);
    return;
#endif
#ifdef AQE_ABL_RESDEV  // (ablation: the result into device memory instead of the pinned host block)
    lean_state_store(T.state, st);
    *reinterpret_cast<aqe_result*>(reinterpret_cast<char*>(T.state) + 256) = res;
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(T.state) + 512), result_check(res, epoch), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
#endif
    // The state is read by a later LAUNCH of the same execution only — the top-up when it is due (launched by fetch(), or
    // already enqueued behind this launch: keep_state), the remaining rounds of a head form — and by the stepwise forms,
    // which do not come here: otherwise its fourteen stores stay home (a store instruction costs one lane what it costs
    // sixty-four: ~25 ns each on this path).
    if (due || goes_on || T.keep_state || !with_result) lean_state_store(T.state, st);
    if (with_result) {
        // The result leaves as ONE store instruction: the finishing lane parks its words in LDS (cheap), lane i of the wave
        // sends word i.  The host polls the pinned result instead of waiting for the end of the launch: the check word
        // tells it when every field has landed (kernels.hpp, result_check).
        static_assert(sizeof(aqe_result) % 8 == 0 && sizeof(aqe_result) / 8 < 63, "one word per lane, the check word after them");
        *reinterpret_cast<aqe_result*>(res_words) = res;
        res_words[sizeof(aqe_result) / 8] = result_check(res, epoch);
    }
#ifdef AQE_LEAN_STAMPS
    g_lean_stamps[static_cast<size_t>(kMaxPersistGrid) * kPersistWaves * 8 + 2] = __builtin_amdgcn_s_memrealtime();
#endif
}

// (all lanes of wave 0, after lean_judge: lane i sends word i of the result, lane sizeof/8 the check word)
__device__ __forceinline__ void lean_send_result(const LeanTail& T, const unsigned long long* res_words, unsigned lane) {
#if defined(AQE_ABL_NOSTORE) || defined(AQE_ABL_RESDEV)
    return;
#endif
    if (T.totals_only || !T.finalize_here) return;
    constexpr unsigned kWords = sizeof(aqe_result) / 8;
    if (lane < kWords) __hip_atomic_store(reinterpret_cast<unsigned long long*>(T.result) + lane, res_words[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else if (lane == kWords) __hip_atomic_store(T.result_seq, res_words[kWords], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// One query on the workgroups bid = 0 .. G-1 (a launch of its own, or one group of a batch's launch).  `a`: the fields
// the sweep needs, in registers; K: the descriptor (the kernel arguments of a single launch, the batch's table in device
// memory otherwise) — read twice: a lane's two runs, and the tail.
template <bool kNT, bool kWide>
__device__ __forceinline__ void lean_query(const LeanLaunch& a, const LeanRuns* const runs, const LeanKarg K, const unsigned bid, const unsigned G, const unsigned long long epoch) {
    // a wave's sums of a round (zero where it swept none); in the folding workgroup, later, the launch's whole partial list
    __shared__ double lds_part[kMaxPersistRounds][kPersistWaves][kVec];
    static_assert(kMaxPersistRounds * kPersistWaves >= kLeanMaxSlots, "the partial list fits where the workgroup's own sums were");
    __shared__ double lds_round[kMaxPersistRounds][kVec];
    __shared__ unsigned lds_slot[kMaxPersistRounds];
    __shared__ unsigned lds_mask[kPersistWaves];  // rounds a wave swept tiles of
    __shared__ u64 lds_tail[64];
    __shared__ unsigned long long lds_res[64];  // the finishing lane's result, word by word (lean_send_result)
    __shared__ int s_last;
    const int lane = threadIdx.x & 63;
    const unsigned wave = threadIdx.x >> 6;
    LEAN_STAMP(0);
    // the run table: lane i holds runs i and i + 64.  One batch of loads, in flight while LDS is cleared.
    // (wide plans: the table goes to LDS instead, one copy per workgroup, and a tile's run is found by bisection)
    __shared__ u64 lds_row0[kWide ? kLeanWideRuns : 1];
    __shared__ unsigned lds_tb[kWide ? kLeanWideRuns : 1], lds_rows[kWide ? kLeanWideRuns : 1], lds_meta[kWide ? kLeanWideRuns : 1], lds_rslot[kWide ? kLeanWideRuns : 1];
    u64 my_row0 = 0, my_row0_hi = 0;
    unsigned my_tb = 0, my_rows = 0, my_meta = 0, my_slot = 0, my_tb_hi = 0, my_rows_hi = 0, my_meta_hi = 0, my_slot_hi = 0;
    if (kWide) {
        for (unsigned i = threadIdx.x; i < a.nruns; i += kPersistThreads) {
            lds_row0[i] = a.wide->row0[i]; lds_tb[i] = a.wide->tile_begin[i]; lds_rows[i] = a.wide->rows[i];
            lds_meta[i] = a.wide->meta[i]; lds_rslot[i] = a.wide->slot[i];
        }
        __syncthreads();
    } else {
        my_row0 = runs->row0[lane]; my_row0_hi = runs->row0[lane + 64];
        my_tb = runs->tile_begin[lane]; my_rows = runs->rows[lane]; my_meta = runs->meta[lane]; my_slot = runs->slot[lane];
        my_tb_hi = runs->tile_begin[lane + 64]; my_rows_hi = runs->rows[lane + 64]; my_meta_hi = runs->meta[lane + 64]; my_slot_hi = runs->slot[lane + 64];
    }
    // The tail of the descriptor — what only the folding workgroup reads — is fetched NOW, by the last wave, one word per
    // lane, and parked in LDS after the sweep: nobody knows yet who will fold, and whoever does must not start fetching
    // descriptor words then, one dependent round trip after the other.
    const bool stager = wave == kPersistWaves - 1u;
    u64 tail_word = 0;
    if (stager && static_cast<unsigned>(lane) < sizeof(LeanTail) / 8u) tail_word = reinterpret_cast<const AQE_KARG u64*>(&K->tail)[lane];
    // every wave clears its own rows of lds_part: nothing to wait for before the sweep
#pragma unroll
    for (unsigned i = 0; i < kMaxPersistRounds * kVec / 64; ++i) {
        const unsigned x = static_cast<unsigned>(lane) + 64u * i;
        lds_part[x >> 3][wave][x & 7u] = 0.0;
    }
    if (a.tail.want_ticks && bid == 0 && threadIdx.x == 0)
        __hip_atomic_store(lean_t0_word(a.counter), static_cast<u64>(__builtin_amdgcn_s_memrealtime()), AQE_RLX);
#ifdef AQE_LEAN_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LEAN_STAMP(1);
#endif

    // Tiles: a contiguous share per workgroup (its wave j takes the j-th, (j + 16)-th, ... of them: the 16 waves stream one
    // window of the column, and the share holds tiles of one round or of a few consecutive ones) — or, for a plan of ONE
    // round, dealt out wave by wave across the launch (tile t to wave t mod 16 G: every HBM channel busy from the start)
    const bool dealt = a.tiles_per_wg == 0u;
    const unsigned t_lo = dealt ? bid * kPersistWaves : bid * a.tiles_per_wg;
    const unsigned t_end = dealt ? a.ntiles : (t_lo + a.tiles_per_wg < a.ntiles ? t_lo + a.tiles_per_wg : a.ntiles);
    const unsigned t_step = dealt ? G * kPersistWaves : kPersistWaves;
    Acc acc;
    unsigned cur = ~0u, cur_slot = 0, touched = 0;  // the round this wave is in; the rounds it has been in
    auto flush = [&]() {
        const double v[7] = {static_cast<double>(acc.na), acc.sa, acc.qa, static_cast<double>(acc.nb), acc.sb, acc.qb, static_cast<double>(acc.nv)};
        const double mine = wave_sum7(v, lane);  // lane 8c holds component c
        if ((lane & 7) == 0 && lane < 56) lds_part[cur][wave][lane >> 3] = mine;
        if (lane == 0) lds_slot[cur] = cur_slot;
        touched |= 1u << cur;
    };
    for (unsigned t = __builtin_amdgcn_readfirstlane(t_lo + wave); t < t_end; t += t_step) {
        // the run that owns tile t: the last whose first tile is <= t (ascending; 0xffffffff past the table)
        unsigned meta, run_slot, run_tb, run_rows, first;
        u64 run_row0;
        if (kWide) {
            unsigned lo = 0, hi = a.nruns;
            while (hi - lo > 1u) {
                const unsigned mid = (lo + hi) >> 1;
                if (static_cast<unsigned>(__builtin_amdgcn_readfirstlane(lds_tb[mid])) <= t) lo = mid; else hi = mid;
            }
            meta = __builtin_amdgcn_readfirstlane(lds_meta[lo]); run_slot = __builtin_amdgcn_readfirstlane(lds_rslot[lo]);
            run_tb = __builtin_amdgcn_readfirstlane(lds_tb[lo]); run_rows = __builtin_amdgcn_readfirstlane(lds_rows[lo]);
            run_row0 = uniform64(lds_row0[lo]);
            first = (t - run_tb) * static_cast<unsigned>(kDenseTileOrdinals);
        } else {
            const unsigned below = static_cast<unsigned>(__builtin_popcountll(__ballot(my_tb <= t))) + static_cast<unsigned>(__builtin_popcountll(__ballot(my_tb_hi <= t)));
            const bool hi = below > 64u;           // wave-uniform: the run sits in the lanes' second set
            const unsigned i = (below - 1u) & 63u;
            meta = hi ? __builtin_amdgcn_readlane(my_meta_hi, i) : __builtin_amdgcn_readlane(my_meta, i);
            run_slot = hi ? __builtin_amdgcn_readlane(my_slot_hi, i) : __builtin_amdgcn_readlane(my_slot, i);
            run_tb = hi ? __builtin_amdgcn_readlane(my_tb_hi, i) : __builtin_amdgcn_readlane(my_tb, i);
            run_rows = hi ? __builtin_amdgcn_readlane(my_rows_hi, i) : __builtin_amdgcn_readlane(my_rows, i);
            run_row0 = hi ? read_lane_u64(my_row0_hi, i) : read_lane_u64(my_row0, i);
            first = (t - run_tb) * static_cast<unsigned>(kDenseTileOrdinals);
            if (meta & kLeanMetaSeg) {  // a row of equal blocks: the tile's block and its place in it (geometry in entry i + 64)
                const unsigned tps = __builtin_amdgcn_readlane(my_meta_hi, i), rel = t - run_tb;
                const unsigned blk = tps == 1u ? rel : rel / tps;
                run_row0 += static_cast<u64>(blk) * read_lane_u64(my_row0_hi, i);
                first = (rel - blk * tps) * static_cast<unsigned>(kDenseTileOrdinals);
            }
        }
        const unsigned r = meta & 0xffu;
        if (r != cur) {
            if (cur != ~0u) { flush(); acc = Acc{}; }
            cur = r;
            cur_slot = run_slot;
        }
        const unsigned rem = run_rows - first;
        const double* const base = a.amount + (run_row0 + first);
        TileAcc ta;
#ifndef AQE_ABL_NOSWEEP  // (ablation builds, tools/ab_ablate.sh: what each stage of a launch costs; results are then wrong)
        lean_tile<kNT>(base, rem, a.amount, lane, a.has_where, a.wmin, a.wmax, a.shift, ta);
#else
        ta.s = static_cast<double>(reinterpret_cast<uintptr_t>(base) & 1u) + rem;
#endif
        merge_tile(acc, ta, (meta & kLeanMetaGroupB) != 0);
#ifdef AQE_LEAN_STAMPS
        if (t == t_lo + wave) LEAN_STAMP(2);
#endif
    }
    LEAN_STAMP(3);
    if (cur != ~0u) flush();
    if (lane == 0) lds_mask[wave] = touched;
    if (stager) lds_tail[lane] = tail_word;
    LEAN_STAMP(4);
    __syncthreads();
#ifdef AQE_ABL_NOTICKET
    return;
#endif

    // ---- the workgroup's partial of every round it swept tiles of: data, drain, ticket (cdna_hip_programming.md G16) ----
    if (wave == 0) {
        unsigned m = 0;
#pragma unroll
        for (unsigned j = 0; j < kPersistWaves; ++j) m |= lds_mask[j];
        m = __builtin_amdgcn_readfirstlane(m);
        while (m) {
            const unsigned r = static_cast<unsigned>(__builtin_ctz(m));
            m &= m - 1u;
            const unsigned info = lds_slot[r];
            const unsigned slot = (info & 0xffffu) + bid - (info >> 16);  // the round's workgroups are consecutive
            if (lane < 7) {
                double x[kPersistWaves];
#pragma unroll
                for (unsigned j = 0; j < kPersistWaves; ++j) x[j] = lds_part[r][j][lane];  // all reads in flight at once
                double s = 0.0;
#pragma unroll
                for (unsigned j = 0; j < kPersistWaves; ++j) s += x[j];  // wave order
                __hip_atomic_store(a.partials + static_cast<size_t>(slot) * kVec + lane, s, AQE_RLX);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the partials are out before the ticket is drawn
        LEAN_STAMP(5);
        if (lane == 0) {
            int last = 0;
            unsigned* const ct = a.counter + static_cast<size_t>(kShards) * kShardStride;
            // at most 256 arrivals: 16 shards of 16 (a same-address atomic serialises at ~16 ns), then the top counter
            constexpr unsigned kLeanShards = 16;
            if (G <= kLeanShards) {
                if (__hip_atomic_fetch_add(ct, 1u, AQE_RLX) == G - 1u) { __hip_atomic_store(ct, 0u, AQE_RLX); last = 1; }
            } else {
                const unsigned sh = bid % kLeanShards;
                const unsigned members = (G - sh + kLeanShards - 1u) / kLeanShards;
                unsigned* const cs = a.counter + static_cast<size_t>(sh) * kShardStride;
                if (__hip_atomic_fetch_add(cs, 1u, AQE_RLX) == members - 1u) {
                    __hip_atomic_store(cs, 0u, AQE_RLX);
                    if (__hip_atomic_fetch_add(ct, 1u, AQE_RLX) == kLeanShards - 1u) { __hip_atomic_store(ct, 0u, AQE_RLX); last = 1; }
                }
            }
            s_last = last;
        }
        LEAN_STAMP(6);
    }
    __syncthreads();
    if (!s_last) return;
#ifdef AQE_ABL_NOFOLD
    return;
#endif

    // ---- the last workgroup to arrive folds the launch.  Everything it needs from the descriptor is in LDS already.  The
    //      whole partial list — fewer than workgroups + rounds slots, whatever the size of the sweep — is fetched in ONE batch
    //      of coalesced loads (three words per thread at most) into the space the workgroup's own sums occupied ----
    const LeanTail& T = *reinterpret_cast<const LeanTail*>(lds_tail);
    const unsigned rounds = T.rounds;
    const unsigned long long t0 = T.want_ticks ? __hip_atomic_load(lean_t0_word(a.counter), AQE_RLX) : 0ull;  // (in flight beside the partials)
    // W waves per round (a power of two; one when the plan has more than eight rounds).  Thread (part, c) of a round's waves
    // sums component c of every (8 W)-th slot of the round straight out of the partial list — fewer than workgroups + rounds
    // slots, whatever the size of the sweep: all of a thread's loads in flight together, ONE round trip for the launch — in
    // ascending order; the eight parts of a wave combine by DPP, the W waves of a round through LDS, in wave order: a fixed
    // order whatever arrives when — bit-reproducible.
#ifdef AQE_ABL_FOLD0  // (ablation: the folding workgroup stops here — tail in LDS read, nothing fetched yet)
    if (rounds != 0x7fffffffu) return;
#endif
    unsigned W = 1;
    while (2u * W * rounds <= kPersistWaves) W *= 2u;
    const unsigned wshift = static_cast<unsigned>(__builtin_ctz(W)), per_pass = kPersistWaves >> wshift;
    for (unsigned q0 = 0; q0 < rounds; q0 += per_pass) {
        const unsigned q = q0 + (wave >> wshift), wsub = wave & (W - 1u);
        const unsigned c = (static_cast<unsigned>(lane) >> 3) & 7u, part = (static_cast<unsigned>(lane) & 7u) + 8u * wsub;
        const bool mine = q < rounds && c < 7u;
        const unsigned b = mine ? T.slot_begin[q] : 0u, e = mine ? T.slot_begin[q + 1u] : 0u;
        double s = 0.0;
        constexpr unsigned kInFlight = 6;  // covers a round of 96 W slots in one turn (a 10 M-row query's rounds: ~65 slots, W = 2)
        for (unsigned sl0 = b + part; __ballot(sl0 < e) != 0; sl0 += 8u * W * kInFlight) {
            double x[kInFlight];
#pragma unroll
            for (unsigned i = 0; i < kInFlight; ++i) {
                const unsigned sl = sl0 + 8u * W * i;
#ifndef AQE_ABL_NOGATHER
                x[i] = sl < e ? __hip_atomic_load(a.partials + static_cast<size_t>(sl) * kVec + c, AQE_RLX) : 0.0;
#else
                x[i] = 1.0;
#endif
            }
#pragma unroll
            for (unsigned i = 0; i < kInFlight; ++i) s += x[i];
        }
        s += dpp_f64<0xB1>(s);   // lane ^ 1
        s += dpp_f64<0x4E>(s);   // lane ^ 2
        s += dpp_f64<0x141>(s);  // the other quad of the eight
        if (mine && (lane & 7) == 0) lds_round[q * W + wsub][c] = s;
    }
#ifdef AQE_ABL_FOLD1  // (ablation: ... here — partials fetched and summed per round, before the barrier)
    if (rounds != 0x7fffffffu) return;
#endif
    __syncthreads();
    LEAN_STAMP_FOLD(1);
    if (wave != 0) return;
#ifdef AQE_ABL_FOLD2  // (ablation: ... here — after the barrier, before the scan and the rules)
    if (rounds != 0x7fffffffu) return;
#endif
    // lane q: the moments through round q (a slot's own total in the totals form, and for the top-up slot)
    const bool tslot = T.topup_slot != 0;
    const bool own = T.totals_only != 0 || (tslot && static_cast<unsigned>(lane) == rounds - 1u);
    // Row L of lds_round (round-major, the W waves of a round in wave order; at most 32 rows) goes to lane L — ONE batch of
    // LDS reads — and the rounds are scanned ACROSS the lanes with DPP shifts: an inclusive prefix over the rows (what the
    // rules judge: the moments through round q) and, beside it, every round's own total (the totals form; the top-up slot).
    // The loop this replaces — every lane walking all rows, one LDS round trip after the other — was 1.4 us of a 10 us
    // launch (profiles/round3_lean_ablation.txt).  Fixed shift pattern: bit-reproducible.
    double tot[7];
    {
        const unsigned nrow = rounds * W;
        double own_t[7];
#pragma unroll
        for (int cc = 0; cc < 7; ++cc) tot[cc] = static_cast<unsigned>(lane) < nrow ? lds_round[lane][cc] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 7; ++cc) {
            double o = tot[cc];  // the W rows of a round: aligned groups of W lanes (W = 1, 2, 4, 8 or 16), butterfly
            if (W >= 2u) o += dpp_f64<0xB1>(o);    // lane ^ 1
            if (W >= 4u) o += dpp_f64<0x4E>(o);    // lane ^ 2
            if (W >= 8u) o += dpp_f64<0x141>(o);   // the other quad of the eight (all four of its lanes agree)
            if (W >= 16u) o += dpp_f64<0x128>(o);  // the other half of the row of sixteen
            own_t[cc] = o;
            double p = tot[cc];  // inclusive prefix within the row of sixteen: shifts by 1, 2, 4, 8 (lanes shifted in from outside add 0)
            p += dpp_f64<0x111>(p);
            p += dpp_f64<0x112>(p);
            p += dpp_f64<0x114>(p);
            p += dpp_f64<0x118>(p);
            const double carry = read_lane_f64(p, 15);  // rows 16 .. 31 sit in the second row of lanes: plus the first row's total
            tot[cc] = p + (lane >= 16 ? carry : 0.0);
        }
        if (W > 1u) {  // lane q takes what the last row of round q holds
            const int src = static_cast<int>((static_cast<unsigned>(lane) + 1u) * W - 1u) & 63;
#pragma unroll
            for (int cc = 0; cc < 7; ++cc) { tot[cc] = __shfl(tot[cc], src, 64); own_t[cc] = __shfl(own_t[cc], src, 64); }
        }
#pragma unroll
        for (int cc = 0; cc < 7; ++cc) tot[cc] = own ? own_t[cc] : tot[cc];
    }
#ifdef AQE_ABL_FOLD3  // (ablation: ... here — rounds scanned, nothing judged or stored)
    if (tot[0] + tot[6] != -1.0) { if (lane == 0) T.state->n_a = tot[0] + tot[1] + tot[2] + tot[3] + tot[4] + tot[5] + tot[6]; return; }
#endif
    lean_judge(T, tot, static_cast<unsigned>(lane), t0, epoch, lds_res);
    wave_lds_handoff();  // (the finishing lane's words, read by the whole wave)
    lean_send_result(T, lds_res, static_cast<unsigned>(lane));
}

template <bool kNT, bool kWide>
__global__ __launch_bounds__(kPersistThreads) void k_sweep_lean(LeanLaunch a) {
    const LeanKarg K = (LeanKarg)__builtin_amdgcn_kernarg_segment_ptr();
    // (the run table is read out of the kernel-argument segment itself, per lane: ordinary global memory)
    lean_query<kNT, kWide>(a, &((const LeanLaunch*)K)->runs, K, blockIdx.x, gridDim.x, a.epoch);
}

// A BATCH of queries in one launch (as k_sweep_multi, persist.hip): the grid is cut into one group of workgroups per
// query, wg_map[blockIdx.x] = query << 32 | group size << 16 | index in the group, group q runs table[q] exactly as a
// launch of its own would on that many workgroups.  Nothing here waits for anything — the last workgroup of a group to
// arrive finishes its query — so groups may be dispatched in any order and there may be more of them than compute units.
template <bool kNT>
__global__ __launch_bounds__(kPersistThreads) void k_sweep_lean_multi(const LeanLaunch* table, const unsigned long long* wg_map, unsigned long long epoch) {
    const u64 me = uniform64(wg_map[blockIdx.x]);
    const LeanKarg K = (LeanKarg)(table + (me >> 32));
    LeanLaunch a;  // what the sweep reads, out of the table once
    a.amount = K->amount; a.ntiles = K->ntiles; a.tiles_per_wg = K->tiles_per_wg;
    a.has_where = K->has_where; a.wmin = K->wmin; a.wmax = K->wmax; a.shift = K->shift;
    a.partials = K->partials; a.counter = K->counter; a.tail.want_ticks = 0; a.wide = nullptr; a.nruns = 0;
    lean_query<kNT, false>(a, &(table + (me >> 32))->runs, K, static_cast<unsigned>(me) & 0xffffu, static_cast<unsigned>(me >> 16) & 0xffffu, epoch);
}

}  // namespace

hipError_t launch_sweep_lean_multi(const LeanLaunch* table, const unsigned long long* wg_map, unsigned long long epoch, unsigned grid, bool nt,
                                   hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    if (nt) {
        if (ev0) hipExtLaunchKernelGGL(k_sweep_lean_multi<true>, dim3(grid), dim3(kPersistThreads), 0, s, ev0, ev1, 0, table, wg_map, epoch);
        else hipLaunchKernelGGL(k_sweep_lean_multi<true>, dim3(grid), dim3(kPersistThreads), 0, s, table, wg_map, epoch);
    } else {
        if (ev0) hipExtLaunchKernelGGL(k_sweep_lean_multi<false>, dim3(grid), dim3(kPersistThreads), 0, s, ev0, ev1, 0, table, wg_map, epoch);
        else hipLaunchKernelGGL(k_sweep_lean_multi<false>, dim3(grid), dim3(kPersistThreads), 0, s, table, wg_map, epoch);
    }
    return hipGetLastError();
}

hipError_t launch_sweep_lean(const LeanLaunch& a, unsigned grid, bool nt, hipStream_t s, hipEvent_t ev0, hipEvent_t ev1) {
    auto go = [&](auto kernel) {
        if (ev0) hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(kPersistThreads), 0, s, ev0, ev1, 0, a);
        else hipLaunchKernelGGL(kernel, dim3(grid), dim3(kPersistThreads), 0, s, a);
    };
    if (a.wide) { if (nt) go(k_sweep_lean<true, true>); else go(k_sweep_lean<false, true>); }
    else { if (nt) go(k_sweep_lean<true, false>); else go(k_sweep_lean<false, false>); }
    return hipGetLastError();
}

#ifdef AQE_LEAN_STAMPS
extern "C" __attribute__((visibility("default"))) int aqe_debug_lean_stamps(unsigned long long* out, size_t words) {
    const size_t all = (kMaxPersistGrid * kPersistWaves + 1) * 8;
    return static_cast<int>(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lean_stamps), 8 * (words < all ? words : all), 0, hipMemcpyDeviceToHost));
}
#endif

}  // namespace aqe
