// host.hpp — what the host-side translation units of libaqe_hip.so share: the context, the plan, and the internal
// functions behind the C ABI of include/aqe_hip.h.  (table.hip: the table in HBM — staging, files, key columns;
// plans.hip: planned queries — forms, launches, fetch, batches; capi.hip: the remaining entry points.)
#pragma once

#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "planner.hpp"

static_assert(sizeof(aqe_record) == 32, "row layout of DB.hpp:17-27");
static_assert(sizeof(aqe::QueryState) % 8 == 0, "state is memset as a block");

namespace aqe {

struct LaunchDesc {
    size_t fam_offset = 0;
    uint32_t nfam = 0;
    uint64_t ntiles = 0;
    uint64_t samples = 0;  // ordinals in this launch's windows (upper bound for the top-up)
};

constexpr size_t kStageChunkRows = 1u << 19;  // 512 Ki rows: 16 MiB of AoS per pinned buffer
constexpr int kStageRing = 4;                 // pinned buffers of the staging ring (kept by the context between stagings)
constexpr size_t kMaxStrideViews = 8;  // stride-major copies of the column a table may hold whatever their size; more while they fit the view budget (table.hip)
constexpr size_t kBatchLanes = 3;  // side streams of the batched multi-GPU form (see ensure_lanes)
// MI355X: a sweep beyond the Infinity Cache is streamed with non-temporal loads (+9-15 % on 0.3-8 GB scans).  Below it the
// two load flavours trade places with how the launches are issued (profiles/round3_nt_policy.txt: on the stream the plan
// warmed up on, non-temporal loads are up to 11 % faster from 48 MB; on a stream the plan has not run on before they cost
// 1.5-2 us per launch at every size) — so the switch stays where both agree.  AQE_NT=0/1 forces it per plan (diagnostics).
constexpr size_t kInfinityCacheBytes = 256ull << 20;
constexpr size_t kGraphMinRounds = 4, kGraphMaxRounds = 8192;  // one-launch-per-round plans replayed as a HIP graph

}  // namespace aqe

// What every plan needs on the device besides its families: workgroup partials, arrival counters, the query state, the
// pinned result block, two events.  Allocating these costs more than planning and running a query (hipHostMalloc
// alone ~100 us), so a destroyed plan hands them back to its context and the next plan takes them over as they are:
// the arrival counters reset themselves, the first launch of a query overwrites the state, partials are written
// before they are read, and the result block's check word carries a launch epoch that is never reused.
constexpr size_t kPoolFams = 64;
struct PlanScratch {
    double* partials = nullptr;
    unsigned* counter = nullptr;
    aqe::QueryState* d_state = nullptr;
    aqe_result* h_result = nullptr;
    aqe_result* d_result = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    aqe::PersistCtl* d_ctl = nullptr;  // persistent sweep: the stop word (tagged with the launch epoch: never reset)
    void* d_rehearsal = nullptr;       // ... and where the monitor's rehearsal writes
    aqe::DevFamily* d_fams_small = nullptr;  // room for a family table of up to kPoolFams entries
};

// One stride-major copy of the column (and, for GROUP BY, of the key columns in the same slot order).
struct StrideView {
    double* amount = nullptr;
    int32_t* keys[2] = {nullptr, nullptr};  // [AQE_GROUP_REGION - 1], [AQE_GROUP_PRODUCT - 1], built on first use
    uint64_t M = 0, q0 = 0;   // slot(r) = (r % step) * M + (r / step - q0)
    uint64_t bytes = 0;       // HBM this view holds (amount + key views)
    uint32_t refs = 0;        // plans laid out over it
    uint32_t cache_refs = 0;  // ... of which the reduce cache's own (evictable with the view)
    uint64_t last_use = 0;
};

// Pinned bounce buffers of the staging path: allocated on the first staging (pinning memory is the most expensive step
// of a small load: ~0.1 ms per MiB) and kept with the context, so that restaging and sharded loads pay it once.
struct StageRing {
    void* buf[aqe::kStageRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t done[aqe::kStageRing] = {nullptr, nullptr, nullptr, nullptr};
    size_t bytes_each = 0;
};

struct aqe_ctx {
    StageRing ring;
    aqe_stage_stats stage_stats{};  // of the most recent staging (aqe_last_stage_stats)
    std::vector<PlanScratch> scratch_pool;
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;
    // table (one shard)
    double* amount = nullptr;
    aqe_record* aos = nullptr;
    bool owns_table = true;
    bool staged = false;
    // lazily built, per table: zone variances of adaptive_block_sample, amount-sorted column of stratified_block_sample
    bool zone_var_valid = false;
    double zone_var[10] = {0};
    double* sorted_amount = nullptr;
    uint32_t* sorted_row = nullptr;
    // stride-major views of the amount column, one per pointer step in use (table.hip ensure_stride_view): at most
    // kMaxStrideViews per table; past that the least recently used view no live plan of the caller holds is evicted
    // (plans of the reduce cache that use it go with it), and when every view is held the new step is swept in place
    std::map<uint64_t, StrideView> stride_views;
    uint64_t view_clock = 0;       // last_use stamps
    uint64_t view_bytes = 0;       // HBM held by views (amount + key views), part of hbm_bytes
    uint64_t view_evictions = 0;   // views dropped to make room, since the table was staged
    uint64_t view_fallbacks = 0;   // plans that wanted a view and were laid out in place instead
    // GROUP BY: key columns (SoA int32), extracted from the AoS rows or generated for a synthetic table on first use
    int32_t* keycol[2] = {nullptr, nullptr};  // [AQE_GROUP_REGION - 1], [AQE_GROUP_PRODUCT - 1]
    int32_t key_min[2] = {0, 0}, key_max[2] = {-1, -1};
    bool synthetic = false;  // made by aqe_generate_synthetic: keys follow from the row number
    std::vector<hipStream_t> lanes;         // side streams of the batched multi-GPU form (aqe_batch), made on first use
    double* grp_partial = nullptr;          // GROUP BY scratch, grown on demand and kept with the context
    size_t grp_partial_bytes = 0;
    aqe_group_result* grp_out = nullptr;    // [aqe::kMaxGroupBins]: device address of grp_out_host
    aqe_group_result* grp_out_host = nullptr;  // pinned, mapped: the finishing kernel writes the groups here
    // the fused single-GPU form (grouped.hip, grouped_fused_epilogue): accumulator + tickets in device memory (zero between
    // launches), a check word per bin beside the pinned groups — what aqe_reduce_grouped polls instead of waiting for the stream
    double* grp_acc = nullptr;
    unsigned* grp_ticket = nullptr;
    unsigned long long* grp_check = nullptr;
    unsigned long long* grp_check_host = nullptr;
    bool ids_dense = false;  // id == first_id + row for every row (detected at staging): key bounds are arithmetic
    int64_t first_id = 0;
    bool dense16 = true;  // dense families may use 16-byte loads (tile sizes depend on it: fixed per table)
    uint64_t n_global = 0, shard_lo = 0, n_local = 0;
    double shift = 0.0;
    double head_cv = 0.0;  // coefficient of variation of the table's first rows (0: unknown): predicts where a CLT query stops
    uint64_t hbm_bytes = 0;
    uint64_t table_epoch = 0;
    // persistent sweep (persist.hip): fixed grid of one 16-wave workgroup per CU (power of two)
    unsigned persist_grid = 0;
    unsigned long long* d_stamps = nullptr;  // diagnostics (env AQE_PERSIST_STAMPS)
    unsigned long long epoch = 1;
    // prepared plans of aqe_reduce / aqe_gather, keyed by the query bytes
    std::vector<std::pair<aqe_query, aqe_plan*>> cache;
};

// One persistent-sweep form of a plan's rounds (persist.hip): the tile list of all slots, who owns tiles
// of which slot, and the workgroup-partial buffer.
struct SweepForm {
    bool ok = false;
    uint32_t slots = 0;       // rounds (+ the top-up as an extra slot in the totals form)
    std::vector<aqe::DevFamily> h_fams;
    aqe::DevFamily* d_fams = nullptr;
    double* d_ppart = nullptr;  // flat workgroup partials: [step_begin[slots] + kDecSteps][8][aqe::kVec]
    std::vector<uint64_t> h_init;  // what d_ppart's block starts out as, until the form is first launched (plans.hip, materialize_form)
    size_t ppart_words = 0;
    uint32_t step_begin[aqe::kMaxPersistRounds + 1] = {0};
    uint64_t round_begin[aqe::kMaxPersistRounds + 1] = {0};
    uint32_t round_mod[aqe::kMaxPersistRounds + 1] = {0};
    uint32_t part_first[aqe::kMaxPersistRounds] = {0}, part_count[aqe::kMaxPersistRounds] = {0};
    uint64_t ntiles = 0, samples = 0;
    uint32_t grid = 0;        // workgroups of the launch (a power of two)
    uint32_t more_rounds = 0; // the form ends before the plan's last round (the head form)
    uint32_t topup_slot = 0;  // the last slot is the plan's top-up
    // the lean variant of the form (lean.hip, k_sweep_lean), when the plan qualifies: its own tile list (runs), its own
    // unpadded slot list; d_ppart then holds [slots][kVec] partials
    bool lean = false;
    uint32_t tiles_per_wg = 0;  // workgroup b owns the tiles [b tiles_per_wg, (b + 1) tiles_per_wg)
    aqe::LeanRuns h_runs{};     // the run table, copied into every launch descriptor
    bool wide = false;          // more runs than that: the table in device memory (behind the partials), aqe::LeanWideRuns
    aqe::LeanWideRuns* d_wide = nullptr;
    uint32_t nruns = 0;
    uint32_t slot_begin[aqe::kMaxPersistRounds + 1] = {0};
};

struct aqe_plan {
    aqe_ctx* ctx = nullptr;
    aqe_query q{};
    aqe::HostPlan host;
    uint64_t table_epoch = 0;
    aqe::DevFamily* d_fams = nullptr;
    aqe::DevFamily* d_fams_small = nullptr;  // the pooled table (d_fams points here when the plan's table fits)
    std::vector<aqe::DevFamily> h_fams;
    std::vector<aqe::LaunchDesc> rounds;
    aqe::LaunchDesc topup;
    uint64_t* d_idx = nullptr;
    aqe::QueryState* d_state = nullptr;
    // The result lives in pinned host memory mapped into the device: the kernel that finishes the query stores the
    // 120 bytes across PCIe itself, and fetching is a stream synchronisation — no copy to enqueue.
    aqe_result* h_result = nullptr;  // pinned, mapped
    aqe_result* d_result = nullptr;  // the device's address of h_result
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    bool tick_timed = false;    // the timed execution was timed by the device clock (result.kernel_ms), not by events
    bool want_ticks = false;    // the launches being enqueued note the device clock
    // Scratch of the hand-off protocols.  It belongs to the plan, not the context, so several plans can be
    // in flight on different streams of one GPU (the tail of one query overlaps the sweep of the next).
    double* partials = nullptr;   // [kMaxBlocks][aqe::kVec]   k_round / k_indexed
    unsigned* counter = nullptr;  // sharded tickets, zero between launches
    aqe::PersistCtl* d_ctl = nullptr;  // persistent sweep: the stop word
    void* d_rehearsal = nullptr;  // persistent sweep: target of the monitor's rehearsal stores
    // Persistent single-launch forms (persist.hip).  `decide`: whole table on this GPU, decisions taken in the
    // kernel (should_stop).  `totals`: any shard, every round plus the top-up swept speculatively, one total per
    // slot written out — the multi-GPU form: ONE all-reduce of the slot totals, then k_replay decides.
    bool persist = false;
    SweepForm decide, totals;
    SweepForm decide_lean, totals_lean, head_lean;  // their lean variants (lean.hip), ok when the plan qualifies
    SweepForm single_lean;                          // a plan of ONE round as a lean launch (runs and rows of blocks; tiles dealt out wave by wave)
    uint64_t last_samples = 0;                      // samples the most recent single launch swept
    bool last_topup_swept = false;                  // ... and whether it swept the top-up along (as one more slot)
    hipGraphExec_t round_graph = nullptr;  // one-launch-per-round form: the launches, captured once
    const double* view_rounds = nullptr;  // stride-major view the rounds' families index (nullptr: the column itself)
    const double* view_topup = nullptr;   // ... and the top-up's
    uint64_t view_step_rounds = 0, view_step_topup = 0;  // the steps of those views (0: none): the plan holds a reference on each
    bool cached = false;                  // owned by the context's reduce cache
    bool nt = false;                      // one execution sweeps more than the Infinity Cache holds: non-temporal loads
    unsigned grid = 0;          // workgroups of the persistent sweep for this plan (the context's, or half of it)
    SweepForm head;             // the first rounds only, on a few workgroups: the single launch of a query predicted to stop early
    volatile unsigned long long* h_seq = nullptr;  // behind h_result: the epoch of the launch whose result is there
    unsigned long long poll_epoch = 0;  // != 0: the last execution ends with a persistent launch of this epoch: fetch() may poll h_seq
    uint32_t last_grid = 0;     // workgroups of the last persistent launch (diagnostics)
    bool predicted_full = false;  // the rules are predicted not to hold before the plan's last round (or there are none)
    bool per_round = false;     // both forms exist and the query is predicted to stop early: launch round by round
    bool expect_topup = false;  // single-launch form: the last execution needed the top-up -> enqueue its launch up front
    int last_exec = 0;  // which form the most recent execution used: 0 one launch per round, 1 decide, 2 totals
    int last_kernel = 0;  // ... and the kernel that swept it (AQE_KERNEL_*)
    uint32_t last_first_unswept = 0;  // ... and the first round that form (the plan's own, or a batch's group form) did NOT sweep
    size_t r_head = 0;          // rounds the head form covers (per_round plans)
    // optional per-launch timing (aqe_plan_set_profiling): one event pair around every sweep launch
    bool profile = false;
    std::vector<hipEvent_t> lev;
    uint32_t lev_used = 0;
};

namespace aqe {

extern thread_local std::string g_create_error;  // aqe_create's failures have no context to carry the text

int fail(aqe_ctx* c, int code, const std::string& msg);

#define HIPCHK(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(ctx, AQE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

// table.hip
void free_table(aqe_ctx* c);
int alloc_table(aqe_ctx* c, uint64_t n_local, bool keep_aos);
int ensure_keys(aqe_ctx* c, int column);
int ensure_zone_variances(aqe_ctx* c);
int ensure_sorted(aqe_ctx* c);
// The stride-major view of step `step` (built on first use): rows r = row0 + k step are contiguous in it.
// slot(r) = (r % step) * M + (r / step - q0); M and q0 come back with the pointer.
int ensure_stride_view(aqe_ctx* c, uint64_t step, const double** view, uint64_t* M, uint64_t* q0);
void release_stride_view(aqe_ctx* c, uint64_t step, bool cached);  // a plan laid out over the view goes away
int ensure_key_view(aqe_ctx* c, int column, uint64_t step, const int32_t** view);  // the key column in the view's slot order

// The shift c of a query's shifted sums (n, S - c n, sum (x - c)^2): the table's (mean of its head, the same on every
// shard), moved into the WHERE range when the query has one and the table's lies outside it.  Rows that pass
// `wmin <= x <= wmax` are then never further from c than the range is wide — a narrow range far from the table's mean
// (amounts 20 .. 30 of a column around 500) would otherwise rebuild a variance of 0.02 from sums of 475^2 per row and
// keep nine digits of it instead of thirteen.  A function of table and query only: every shard takes the same c.
inline double query_shift(const aqe_ctx* c, const aqe_query& q) {
    double s = c->shift;
    if (q.has_where && q.where_min <= q.where_max) s = std::min(std::max(s, q.where_min), q.where_max);
    return s;
}

// plans.hip
void destroy_plan(aqe_plan* p, bool device_idle = false);  // device_idle: the caller has just synchronised the device
void drop_cache(aqe_ctx* c);
SweepCommon sweep_common(const aqe_plan* p, const DevFamily* fams, uint32_t nfam, bool topup = false);
FoldParams fold_params(const aqe_plan* p, bool topup);
FinalizeParams finalize_params(const aqe_plan* p);
int plan_is_current(aqe_plan* p);
// Families handed in by the caller instead of planned from q.method (aqe_plan_create_families).
struct GivenFamilies {
    const aqe_family* fams;
    uint32_t n;
    uint64_t global_samples;
    bool on_sorted;
};
int create_plan(aqe_ctx* c, const aqe_query* q, aqe_plan** out, const GivenFamilies* given = nullptr);
int cached_plan(aqe_ctx* c, const aqe_query* q, aqe_plan** out);
int enqueue_all(aqe_plan* p, hipStream_t s, bool timed);
int run_sync(aqe_plan* p, hipStream_t s, bool timed);
int fetch(aqe_plan* p, aqe_result* out, hipStream_t s, bool already_synced = false);

}  // namespace aqe
