// kernels.hpp — launch interface between the C-ABI layer (capi.hip) and the gfx950 kernels (kernels.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/aqe_hip.h"
#include "planner.hpp"

namespace aqe {

constexpr int kBlockThreads = 256;          // 4 wave64 per workgroup
constexpr int kWavesPerBlock = kBlockThreads / 64;
constexpr int kTileUnroll = 8;              // loads in flight per lane
constexpr int kTileOrdinals = 64 * kTileUnroll;  // ordinals one wave folds per tile
// Dense families (step 1, single pointer: blocks, pages, exact scans) are swept with 16-byte loads, two
// rows per lane per load: their tiles are twice as long.
constexpr int kDenseTileOrdinals = 2 * kTileOrdinals;
// (short segments — pages of 128 rows — keep the 512-ordinal tile: a long tile would idle most of the wave)
__host__ __device__ inline bool is_dense16(uint64_t step, uint32_t flags, uint64_t seg_len) { return step == 1 && !(flags & AQE_F_PAIR) && seg_len >= static_cast<uint64_t>(kTileOrdinals); }
inline uint64_t tile_ordinals(uint64_t step, uint32_t flags, uint64_t seg_len) { return is_dense16(step, flags, seg_len) ? kDenseTileOrdinals : kTileOrdinals; }
// Internal family flag (beside the ABI's AQE_F_*): short segments (pages of 128 rows) are tiled along the ORDINAL axis,
// a tile spanning several segments, instead of one mostly idle tile per segment.
constexpr uint32_t kFamLinear = 1u << 8;
constexpr int kMaxBlocks = 2048;            // 8 workgroups per CU on 256 CUs
constexpr unsigned kRoundGridCap = 1024;    // k_round: 4 workgroups per CU (launch_round)
constexpr int kVec = AQE_MOMENT_VEC;
// Arrival tickets are sharded: a same-address device atomic costs ~20 ns and serialises, so 2048
// workgroups on one counter would spend 40 us arriving.  Workgroup b draws from shard b % kShards (each on
// its own 128-byte line); the last of a shard draws from the top counter.
constexpr int kShards = 64;
constexpr int kShardStride = 32;  // u32 words between shard counters (128 B)
constexpr int kCounterWords = (kShards + 1) * kShardStride;

// A family as the device sees it: the ABI family plus its tile decomposition.
struct DevFamily {
    uint64_t row0, pitch, seg_len, step;
    uint64_t ord_lo, ord_hi;
    uint64_t tile_begin;     // first tile id (within the launch) owned by this family
    uint64_t seg_lo;         // first segment the window touches
    uint64_t tiles_per_seg;  // 0: the window lies in one segment and tile j starts at ordinal (j_lo+j)*tile
    uint64_t j_lo;
    uint64_t out_begin;      // gather: position of ordinal ord_lo in the output
    uint64_t row0_b, ord_lo_b, ord_hi_b;  // AQE_F_PAIR: the slow pointer sharing this sweep (group 1)
    uint64_t out_begin_b;
    uint32_t group, flags;
    // persistent sweep only: the round (slot) the family belongs to and that round's end tile — a wave learns
    // both with the family it has to load anyway, instead of walking round_begin[] load by dependent load
    uint64_t round_end;
    uint32_t round, pad_;
};

// Running state of one query on the device: moment triples folded round by round, the CLT
// decision, and the should_stop flag later launches test on entry.
struct QueryState {
    // shifted sums (n, sum(x-c), sum (x-c)^2): additive, so folds are exact re-groupings
    double n_a, sd_a, qd_a;    // group a: the CLT leader, fast worker 0 (or every sample of a non-CLT query)
    double n_b, sd_b, qd_b;    // group b: every other CLT worker
    double n_p, sd_p, qd_p;    // pooled a+b, plus the top-up
    double visited;            // samples drawn (>= n_p when a WHERE filter drops some)
    double topup;              // rows added by the top-up
    int32_t stop;              // should_stop: set by the CLT rules, read by every later launch
    int32_t converged;         // 0 none, 1 error rule, 2 cross-validation rule
    int32_t rounds;            // rounds folded
    int32_t error;             // device protocol error (persistent sweep timed out waiting for a decision)
    unsigned long long t0;     // device clock (s_memrealtime, 100 MHz) when the query's first launch started, if it was asked to note it
};

struct FoldParams {
    double shift;      // c of the shifted sums
    double z, e;       // CLT
    int32_t base;      // CLT: int(N*pct/100)
    int32_t is_clt;
    int32_t is_topup;  // this launch is the top-up (folds into the pooled triple only)
    int32_t pad;
};

struct FinalizeParams {
    uint64_t n_global;
    double pct;
    double shift;
    int32_t agg, convention, is_exact, is_clt;
};

// What a tile sweep needs: the column, the family table and the row filter.
struct SweepCommon {
    const double* amount;   // this shard's amount column
    uint64_t shard_lo;
    const DevFamily* fams;
    uint32_t nfam;
    int32_t has_where;
    double wmin, wmax;
    double shift;           // c of the shifted sums
    int32_t dense16;        // 16-byte loads allowed on dense families (rows 0..1 of the column are readable)
    int32_t nt;             // 1: what this launch sweeps is larger than the Infinity Cache: non-temporal loads on the dense path
};

struct RoundLaunch {
    SweepCommon sw;
    uint64_t ntiles;
    double* partials;       // [kMaxBlocks][kVec]
    unsigned* counter;      // arrival ticket, zero between launches
    double* out_vec;        // reduced vector of this launch (may be null)
    QueryState* state;
    int32_t fused;          // last arriver folds into state (single-GPU form)
    int32_t check_stop;     // leave at once when state->stop is set
    int32_t reset_state;    // first launch of a query: fold into a zeroed state (no separate memset)
    int32_t do_finalize;    // last launch of a query (fused form): the folding thread also writes the result
    FoldParams fold;
    FinalizeParams fin;
    aqe_result* result;
    unsigned long long epoch;        // with result_seq: the launch that finishes the query also writes result_check(result, epoch)
    unsigned long long* result_seq;  // beside the (pinned) result, or null
    uint32_t want_ticks;             // 1: time the query on the device clock (first launch's start -> result written) into result.kernel_ms
    uint32_t pad_;
};

// ---- persistent single-launch sweep of a multi-round (CLT) query: persist.hip ------------------
constexpr int kMaxPersistRounds = 32;
constexpr int kPersistThreads = 1024;  // one workgroup of 16 waves per CU
constexpr int kPersistWaves = kPersistThreads / 64;
constexpr int kMaxPersistGrid = 256;   // workgroups (a power of two <= CU count)
constexpr int kPersistInlineFams = 20; // family tables up to this size travel in the kernel arguments (4 KB of them at most)
constexpr int kDecSteps = 32;          // monitor: steps (8 workgroup partials each) per batch of loads

static_assert(kMaxPersistGrid / 8 <= kDecSteps, "a round's slots fit the monitor's window");
constexpr unsigned long long kSlotAlways = ~0ull;  // flag word of a pad slot: always "published"

// The word a persistent launch writes beside its result in pinned host memory: a hash of the launch's epoch and of
// every field of the result.  The stores of the result and of this word may become visible to the host in any order
// (they leave through different memory channels), so the host does not wait for a flag: it reads result and word
// until they agree — which they do exactly when all of this launch's stores have landed.
__host__ __device__ inline unsigned long long result_check(const aqe_result& r, unsigned long long epoch) {
    // Two running sums over the result's words, the second position-weighted (it adds the first after every word), seeded
    // with the epoch: a block in which some words are the previous execution's disagrees in one sum or the other.  It
    // sits on the critical path of every query (the finishing lane computes it before its last store): ~60 instructions,
    // where a multiplicative hash per word took ~170.
    unsigned long long a = epoch * 0x9E3779B97F4A7C15ull + 1ull, b = 0;
    auto mix = [&a, &b](unsigned long long w) { a += w; b += a; };
    auto bits = [](double d) { unsigned long long u; __builtin_memcpy(&u, &d, 8); return u; };
    mix(bits(r.value)); mix(bits(r.ci_lower)); mix(bits(r.ci_upper)); mix(bits(r.margin));
    mix(bits(r.sum)); mix(bits(r.sumsq)); mix(bits(r.mean)); mix(bits(r.m2));
    mix(r.n); mix(r.visited); mix(r.topup);
    mix(static_cast<unsigned long long>(static_cast<uint32_t>(r.converged)) | (static_cast<unsigned long long>(static_cast<uint32_t>(r.rounds)) << 32));
    mix(bits(r.kernel_ms)); mix(r.bytes_algorithmic);
    mix(static_cast<unsigned long long>(static_cast<uint32_t>(r.device_status)) | (static_cast<unsigned long long>(static_cast<uint32_t>(r.topup_pending)) << 32));
    return a ^ ((b << 32) | (b >> 32));
}

// Control block in device memory (zeroed once per plan).
struct PersistCtl {
    unsigned long long stop_word;  // (epoch << 8) | 1 once the monitor has ended the query
    unsigned long long pad[15];
};

struct PersistLaunch {
    SweepCommon sw;            // family table of ALL rounds, tile_begin numbered across the whole launch
    uint64_t ntiles;
    uint64_t round_begin[kMaxPersistRounds + 1];  // first tile of each round; [rounds] == ntiles
    uint32_t round_mod[kMaxPersistRounds + 1];    // round_begin[r] mod (number of sweepers)
    uint32_t part_first[kMaxPersistRounds];       // workgroups owning tiles of round r: the cyclic run
    uint32_t part_count[kMaxPersistRounds];       //   [part_first, part_first + part_count) mod grid
    // Workgroup partials live in ONE flat list in round order: round r owns the slots
    // [8 step_begin[r], 8 step_begin[r+1]) (part_count[r] rounded up to 8; pad slots stay zero and carry the
    // flag kSlotAlways), the i-th workgroup of its run writes slot 8 step_begin[r] + i: doubles 0..6 = its
    // partial, word 7 = the launch's epoch once the partial is out.  A "step" is 8 slots = 64 doubles = one
    // wave load.
    uint32_t step_begin[kMaxPersistRounds + 1];
    uint32_t rounds;
    uint32_t inline_fams;      // 1: use `fams` below (kernel-argument copy of the table)
    uint32_t finalize_here;    // 1: the monitor also writes the result when it ends the query
    uint32_t topup_gate;       // 1: the plan has a top-up stage: mark the result topup_pending when it is due (DB.cpp:1032)
    uint32_t more_rounds;      // 1: the plan has rounds behind this launch's last: running out of rounds here does not end the query
    uint32_t topup_slot;       // 1: the LAST slot is the top-up, swept with the rounds: the monitor judges once everything is in and adds it when due
    uint32_t want_ticks;       // 1: the monitor times the query on the device clock into result.kernel_ms (and state.t0)
    uint32_t totals_only;      // 1: no decisions in the kernel; the monitor writes every round's total to out_totals
    unsigned long long epoch;  // distinguishes this launch's flags from the previous launch's
    PersistCtl* ctl;
    double* partials;          // [step_begin[rounds] + kDecSteps][8][kVec]  workgroup partials, flat
    double* out_totals;        // totals_only: [rounds][kVec] (this shard's slot totals, for the all-reduce)
    QueryState* state;
    FoldParams fold;
    FinalizeParams fin;
    aqe_result* result;
    QueryState* rehearsal_state;     // where the monitor's rehearsal writes (never read)
    aqe_result* rehearsal_result;
    unsigned long long* result_seq;  // beside the result (pinned host memory): receives result_check(result, epoch) — what fetch() polls
    unsigned long long* stamps;  // diagnostics only (AQE_PERSIST_STAMPS): s_memrealtime marks, else null
    // first tile of family i (0xffffffff past the table): a sweeper finds a tile's family by comparing against these
    // — kernel arguments at fixed offsets, in registers after the prologue's one batch of loads; no search through memory
    uint32_t fam_begin[kPersistInlineFams];
    DevFamily fams[kPersistInlineFams];
};

// ev0/ev1 (optional): events that receive the dispatch's own begin/end timestamps (hipExtLaunchKernelGGL)
static_assert(sizeof(PersistLaunch) <= 4096, "kernel arguments are limited to 4 KB");
hipError_t launch_sweep_persist(const PersistLaunch& a, unsigned grid, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// A batch of queries in one launch: table[q] describes query q (as a launch of its own on `group size` workgroups would be
// described; its epoch field is ignored), wg_map[b] = q << 32 | group size << 16 | index within the group for workgroup b.
hipError_t launch_sweep_multi(const PersistLaunch* table, const unsigned long long* wg_map, unsigned long long epoch, unsigned grid, bool nt,
                              hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

// ---- lean single-launch sweep: lean.hip ----------------------------------------------------------------------------
// A RUN is a family in its simplest form: `rows` consecutive rows of the column (or of a stride-major view) that all
// belong to the sample — what every strided pointer read through a view comes to, and what an exact scan is.  A SEGMENTED
// run is a row of equal blocks `pitch` rows apart (block_sample, DB.cpp:1151-1181): tile t of it lies in block
// t / tiles_per_block.
constexpr int kLeanMaxRuns = 128;                                       // two runs per lane of a wave: lane i holds runs i and i + 64
constexpr int kLeanMaxSlots = kMaxPersistGrid + kMaxPersistRounds;      // a round boundary splits at most one workgroup
constexpr uint32_t kLeanMetaGroupB = 1u << 8;   // meta: the run feeds group b (every CLT worker but the leader)
constexpr uint32_t kLeanMetaSeg = 1u << 9;      // meta: segmented run; entry i + 64 holds its geometry (see LeanRuns)
struct LeanRuns {  // part of the launch descriptor, structure of arrays: lane i of every wave holds runs i and i + 64 in registers
    uint64_t row0[kLeanMaxRuns];        // first row, relative to LeanLaunch::amount
    uint32_t tile_begin[kLeanMaxRuns];  // first tile of the run in the launch's tile list; 0xffffffff past the table
    uint32_t rows[kLeanMaxRuns];        // rows of the run (of ONE block for a segmented run)
    uint32_t meta[kLeanMaxRuns];        // round | kLeanMeta*
    uint32_t slot[kLeanMaxRuns];        // of the run's round: its first slot | the first workgroup that sweeps tiles of it << 16
    // A segmented run i (< 64; such plans hold at most 64 runs) keeps its geometry in entry i + 64, whose tile_begin stays
    // 0xffffffff: row0[i + 64] = rows between block starts, meta[i + 64] = tiles per block.
};
// Plans of more runs than the lanes hold (many pointers: T >= 64 on 10 M rows) bring their table in device memory; every
// workgroup copies it to LDS and finds a tile's run by bisection (k_sweep_lean<.., wide>).
constexpr int kLeanWideRuns = 512;
struct LeanWideRuns {
    uint64_t row0[kLeanWideRuns];
    uint32_t tile_begin[kLeanWideRuns], rows[kLeanWideRuns], meta[kLeanWideRuns], slot[kLeanWideRuns];
};
// What only the workgroup that finishes the query needs.  One wave of every workgroup loads it beside its first tile and
// parks it in LDS, so that after the last ticket nothing is fetched from the descriptor any more: each such fetch was a
// dependent round trip on the query's critical path (profiles/round2_lean_timeline.txt: 5 us of tail behind a 3.5 us sweep).
struct LeanTail {
    uint32_t rounds, finalize_here, topup_gate, more_rounds, topup_slot, want_ticks, totals_only;  // as in PersistLaunch
    uint32_t keep_state;       // 1: a launch that reads the query state is already enqueued behind this one (the top-up, device-gated)
    uint32_t slot_begin[kMaxPersistRounds + 1];  // round r owns the slots [slot_begin[r], slot_begin[r + 1]) of the flat list
    uint32_t pad1;
    FoldParams fold;
    FinalizeParams fin;
    double* out_totals;        // totals_only: [rounds][kVec]
    QueryState* state;
    aqe_result* result;
    unsigned long long* result_seq;
};
static_assert(sizeof(LeanTail) % 8 == 0 && sizeof(LeanTail) <= 64 * 8, "one 8-byte load per lane of a wave stages the tail");
struct LeanLaunch {
    const double* amount;
    uint32_t ntiles;
    int32_t has_where;
    uint32_t tiles_per_wg;     // workgroup b owns the tiles [b tiles_per_wg, (b + 1) tiles_per_wg); 0: tiles are dealt out
                               // wave by wave across the whole launch (single-round plans: one slot per workgroup)
    uint32_t nruns;
    double wmin, wmax, shift;
    double* partials;          // [slots][kVec]: doubles 0..6 = a workgroup's partial of one round
    unsigned* counter;         // arrival tickets (k_round's), zero between launches
    unsigned long long epoch;
    const LeanWideRuns* wide;  // null: the run table below; else the plan's table of `nruns` > kLeanMaxRuns runs
    LeanTail tail;
    // The run table travels IN the descriptor — the kernel arguments of a single launch, the batch's table otherwise — so a
    // wave's very first loads (its lane's two runs) depend on nothing but the descriptor's address.
    LeanRuns runs;
};
static_assert(sizeof(LeanLaunch) <= 4096, "kernel arguments are limited to 4 KB");
hipError_t launch_sweep_lean(const LeanLaunch& a, unsigned grid, bool nt, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
// a batch of queries in one launch: table[q] describes query q (its epoch field is ignored), wg_map as for launch_sweep_multi
hipError_t launch_sweep_lean_multi(const LeanLaunch* table, const unsigned long long* wg_map, unsigned long long epoch, unsigned grid, bool nt,
                                   hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);

hipError_t launch_replay(const double* totals, uint32_t rounds, uint32_t has_topup, const FoldParams& fp,
                         const FinalizeParams& fin, QueryState* state, aqe_result* result, hipStream_t s);

// one plan of a batch, for k_replay_batch
struct ReplayItem {
    uint32_t rounds, has_topup;
    FoldParams fp;
    FinalizeParams fin;
    QueryState* state;
    aqe_result* result;
};
hipError_t launch_replay_batch(const ReplayItem* items, uint32_t n, const double* totals, uint64_t row_stride, hipStream_t s);

hipError_t launch_round(const RoundLaunch& a, hipStream_t s, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr);
hipError_t launch_indexed(const RoundLaunch& a, const uint64_t* idx, uint64_t n_idx, hipStream_t s, hipEvent_t ev0 = nullptr,
                          hipEvent_t ev1 = nullptr);
// AQE_M_RANDOM_DEVICE: rows perm.lo + P(k), k < perm.target, drawn in the kernel (planner.hpp PermSpec)
hipError_t launch_permuted(const RoundLaunch& a, const PermSpec& perm, uint64_t shard_lo, uint64_t shard_rows, hipStream_t s, hipEvent_t ev0 = nullptr,
                           hipEvent_t ev1 = nullptr);
hipError_t launch_gather_permuted(const aqe_record* aos, const PermSpec& perm, aqe_record* out, hipStream_t s);
hipError_t launch_update(QueryState* state, const double* vec, const FoldParams& p, int reset_state, hipStream_t s);
hipError_t launch_finalize(const QueryState* state, const FinalizeParams& p, aqe_result* out, hipStream_t s);

hipError_t launch_gather(const aqe_record* aos, uint64_t shard_lo, const DevFamily* fams, uint32_t nfam,
                         uint64_t ntiles, aqe_record* out, int dense16, const uint32_t* perm, hipStream_t s);
// sort.hip: ascending amounts + their rows (rocPRIM radix sort), synchronous
hipError_t sort_amounts(const double* amount, uint64_t n, double* sorted_amount, uint32_t* sorted_row, hipStream_t s);
hipError_t sorted_counts(const double* sorted_amount, uint64_t n, const double* values, uint32_t m, uint64_t* n_less,
                         uint64_t* n_less_equal, hipStream_t s);
hipError_t launch_gather_indexed(const aqe_record* aos, uint64_t shard_lo, const uint64_t* idx, uint64_t n,
                                 aqe_record* out, hipStream_t s);

hipError_t launch_id_bounds(const aqe_record* aos, uint64_t n, int64_t id_min, int64_t id_max, uint64_t* out, hipStream_t s);
hipError_t launch_stride_view(const double* amount, uint64_t n, uint64_t shard_lo, uint64_t step, uint64_t M, uint64_t q0, double* out,
                              hipStream_t s);
hipError_t launch_stride_view_keys(const int32_t* keys, uint64_t n, uint64_t shard_lo, uint64_t step, uint64_t M, uint64_t q0, int32_t* out,
                                   hipStream_t s);
hipError_t launch_split_amount(const aqe_record* aos, double* amount, uint64_t n, hipStream_t s);
hipError_t launch_synth(aqe_record* aos_or_null, double* amount, uint64_t n, uint64_t first_row, uint64_t seed,
                        hipStream_t s);

// ---- GROUP BY: grouped.hip -----------------------------------------------------------------------
constexpr int kMaxGroupBins = 1024;     // key_max - key_min + 1 of a group column
constexpr unsigned kGroupedMaxBlocks = 1024;
hipError_t launch_extract_key(const aqe_record* aos, int32_t* out, uint64_t n, int column, hipStream_t s);
hipError_t launch_synth_key(int32_t* out, uint64_t n, uint64_t first_row, int column, hipStream_t s);
hipError_t launch_key_range(const int32_t* keys, uint64_t n, int32_t* out2, hipStream_t s);
unsigned grouped_grid(uint64_t ntiles);
// the fused single-GPU form of a grouped sweep (grouped.hip, grouped_fused_epilogue)
struct GroupFuse {
    double* acc = nullptr;          // [nbins][4], zero between launches
    unsigned* ticket = nullptr;     // kCounterWords sharded arrival tickets, zero between launches
    aqe_group_result* out = nullptr;       // pinned: every bin's group
    unsigned long long* check = nullptr;   // pinned: group_check(out[b], epoch) per bin — what the host polls
    unsigned long long epoch = 0;
    double shift = 0.0, pct = 0.0;
    int32_t agg = 0, pad = 0;
};
// (as result_check: the stores of a group and of its check word may land in any order; they agree when all have)
__host__ __device__ inline unsigned long long group_check(const aqe_group_result& r, unsigned long long epoch) {
    unsigned long long a = epoch * 0x9E3779B97F4A7C15ull + 1ull, b = 0;
    auto mix = [&a, &b](unsigned long long w) { a += w; b += a; };
    auto bits = [](double d) { unsigned long long u; __builtin_memcpy(&u, &d, 8); return u; };
    mix(static_cast<unsigned long long>(r.key)); mix(r.n); mix(r.visited);
    mix(bits(r.sum)); mix(bits(r.sumsq)); mix(bits(r.mean)); mix(bits(r.value)); mix(bits(r.ci_lower)); mix(bits(r.ci_upper));
    return a ^ ((b << 32) | (b >> 32));
}
hipError_t launch_grouped(const SweepCommon& sw, uint64_t ntiles, const int32_t* keys, int32_t key_min, uint32_t nbins, double* partial,
                          unsigned grid, hipStream_t s, const GroupFuse* fuse = nullptr);
hipError_t launch_grouped_sum(const double* partial, unsigned nblocks, uint32_t nbins, double* bins, hipStream_t s);
hipError_t launch_grouped_sum_finish(const double* partial, unsigned nblocks, uint32_t nbins, int32_t key_min, double shift, double pct, int agg,
                                     aqe_group_result* out, hipStream_t s);
hipError_t launch_grouped_finish(const double* bins, uint32_t nbins, int32_t key_min, double shift, double pct, int agg, aqe_group_result* out,
                                 hipStream_t s);

}  // namespace aqe
