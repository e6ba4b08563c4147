// capi.hip — the C ABI of include/aqe_hip.h: context, staging into HBM, plan objects, enqueue/fetch.
// Host code only (compiled by hipcc for the HIP runtime API); the kernels live in kernels.hip.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.hpp"
#include "planner.hpp"

using namespace aqe;

static_assert(sizeof(aqe_record) == 32, "row layout of DB.hpp:17-27");
static_assert(sizeof(QueryState) % 8 == 0, "state is memset as a block");

namespace {
thread_local std::string g_create_error;

struct LaunchDesc {
    size_t fam_offset = 0;
    uint32_t nfam = 0;
    uint64_t ntiles = 0;
    uint64_t samples = 0;  // ordinals in this launch's windows (upper bound for the top-up)
};

constexpr size_t kStageChunkRows = 1u << 21;  // 2 Mi rows: 64 MiB of AoS per pinned buffer
}  // namespace

constexpr size_t kBatchLanes = 3;  // side streams of the batched multi-GPU form (see ensure_lanes)
constexpr size_t kGraphMinRounds = 4, kGraphMaxRounds = 8192;  // one-launch-per-round plans replayed as a HIP graph

struct aqe_ctx {
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
    // GROUP BY: key columns (SoA int32), extracted from the AoS rows or generated for a synthetic table on first use
    int32_t* keycol[2] = {nullptr, nullptr};  // [AQE_GROUP_REGION - 1], [AQE_GROUP_PRODUCT - 1]
    int32_t key_min[2] = {0, 0}, key_max[2] = {-1, -1};
    bool synthetic = false;  // made by aqe_generate_synthetic: keys follow from the row number
    std::vector<hipStream_t> lanes;         // side streams of the batched multi-GPU form (aqe_batch), made on first use
    double* grp_partial = nullptr;          // GROUP BY scratch, grown on demand and kept with the context
    size_t grp_partial_bytes = 0;
    aqe_group_result* grp_out = nullptr;    // [kMaxGroupBins]
    double* grp_bins = nullptr;             // [kMaxGroupBins][4] (single-GPU form)
    bool ids_dense = false;  // id == first_id + row for every row (detected at staging): key bounds are arithmetic
    int64_t first_id = 0;
    bool dense16 = true;  // dense families may use 16-byte loads (tile sizes depend on it: fixed per table)
    uint64_t n_global = 0, shard_lo = 0, n_local = 0;
    double shift = 0.0;
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
    std::vector<DevFamily> h_fams;
    DevFamily* d_fams = nullptr;
    double* d_ppart = nullptr;  // flat workgroup partials: [step_begin[slots] + kDecSteps][8][kVec]
    uint32_t step_begin[kMaxPersistRounds + 1] = {0};
    uint64_t round_begin[kMaxPersistRounds + 1] = {0};
    uint32_t round_mod[kMaxPersistRounds + 1] = {0};
    uint32_t part_first[kMaxPersistRounds] = {0}, part_count[kMaxPersistRounds] = {0};
    uint64_t ntiles = 0, samples = 0;
};

struct aqe_plan {
    aqe_ctx* ctx = nullptr;
    aqe_query q{};
    HostPlan host;
    uint64_t table_epoch = 0;
    DevFamily* d_fams = nullptr;
    std::vector<DevFamily> h_fams;
    std::vector<LaunchDesc> rounds;
    LaunchDesc topup;
    uint64_t* d_idx = nullptr;
    QueryState* d_state = nullptr;
    // The result lives in pinned host memory mapped into the device: the kernel that finishes the query stores the
    // 120 bytes across PCIe itself, and fetching is a stream synchronisation — no copy to enqueue.
    aqe_result* h_result = nullptr;  // pinned, mapped
    aqe_result* d_result = nullptr;  // the device's address of h_result
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timed = false;
    // Scratch of the hand-off protocols.  It belongs to the plan, not the context, so several plans can be
    // in flight on different streams of one GPU (the tail of one query overlaps the sweep of the next).
    double* partials = nullptr;   // [kMaxBlocks][kVec]   k_round / k_indexed
    unsigned* counter = nullptr;  // sharded tickets, zero between launches
    PersistCtl* d_ctl = nullptr;  // persistent sweep: the stop word
    void* d_rehearsal = nullptr;  // persistent sweep: target of the monitor's rehearsal stores
    // Persistent single-launch forms (persist.hip).  `decide`: whole table on this GPU, decisions taken in the
    // kernel (should_stop).  `totals`: any shard, every round plus the top-up swept speculatively, one total per
    // slot written out — the multi-GPU form: ONE all-reduce of the slot totals, then k_replay decides.
    bool persist = false;
    SweepForm decide, totals;
    hipGraphExec_t round_graph = nullptr;  // one-launch-per-round form: the launches, captured once
    int last_exec = 0;  // which form the most recent execution used: 0 one launch per round, 1 decide, 2 totals
    // optional per-launch timing (aqe_plan_set_profiling): one event pair around every sweep launch
    bool profile = false;
    std::vector<hipEvent_t> lev;
    uint32_t lev_used = 0;
};

namespace {

int fail(aqe_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define HIPCHK(ctx, expr)                                                                          \
    do {                                                                                           \
        hipError_t e__ = (expr);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return fail(ctx, AQE_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));     \
    } while (0)

// Shift c of the shifted moments: the mean of the table's first rows (up to 1024), so that one outlying
// first row cannot push c outside the data's range.  Every shard of a table must use the same value.
constexpr uint64_t kShiftRows = 1024;
double shift_of_rows(const aqe_record* rows, uint64_t n) {
    const uint64_t m = std::min<uint64_t>(n, kShiftRows);
    double s = 0.0;
    for (uint64_t i = 0; i < m; ++i) s += rows[i].amount;
    return m ? s / static_cast<double>(m) : 0.0;
}

void free_table(aqe_ctx* c) {
    if (c->owns_table) {
        if (c->amount) (void)hipFree(c->amount);
        if (c->aos) (void)hipFree(c->aos);
    }
    if (c->sorted_amount) (void)hipFree(c->sorted_amount);
    if (c->sorted_row) (void)hipFree(c->sorted_row);
    for (int k = 0; k < 2; ++k) {
        if (c->keycol[k]) (void)hipFree(c->keycol[k]);
        c->keycol[k] = nullptr;
    }
    c->synthetic = false;
    c->sorted_amount = nullptr;
    c->sorted_row = nullptr;
    c->zone_var_valid = false;
    c->amount = nullptr;
    c->aos = nullptr;
    c->owns_table = true;
    c->staged = false;
    c->ids_dense = false;
    c->first_id = 0;
    c->n_global = c->shard_lo = c->n_local = 0;
    c->hbm_bytes = 0;
    c->table_epoch++;
}

void destroy_plan(aqe_plan* p) {
    if (!p) return;
    if (p->d_fams) (void)hipFree(p->d_fams);
    if (p->d_idx) (void)hipFree(p->d_idx);
    if (p->partials) (void)hipFree(p->partials);
    if (p->counter) (void)hipFree(p->counter);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    if (p->d_rehearsal) (void)hipFree(p->d_rehearsal);
    for (SweepForm* f : {&p->decide, &p->totals}) {
        if (f->d_fams) (void)hipFree(f->d_fams);
        if (f->d_ppart) (void)hipFree(f->d_ppart);
    }
    if (p->d_state) (void)hipFree(p->d_state);
    if (p->round_graph) (void)hipGraphExecDestroy(p->round_graph);
    if (p->h_result) (void)hipHostFree(p->h_result);
    if (p->ev0) (void)hipEventDestroy(p->ev0);
    if (p->ev1) (void)hipEventDestroy(p->ev1);
    for (auto e : p->lev) (void)hipEventDestroy(e);
    delete p;
}

void drop_cache(aqe_ctx* c) {
    for (auto& kv : c->cache) destroy_plan(kv.second);
    c->cache.clear();
}

int alloc_table(aqe_ctx* c, uint64_t n_local, bool keep_aos) {
    free_table(c);
    drop_cache(c);
    if (n_local == 0) return AQE_OK;
    // one spare double behind the column: the 16-byte dense loads park masked lanes on rows 0..1
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->amount), (n_local + 1) * sizeof(double)));
    c->hbm_bytes = (n_local + 1) * sizeof(double);
    c->dense16 = true;
    if (keep_aos) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->aos), n_local * sizeof(aqe_record)));
        c->hbm_bytes += n_local * sizeof(aqe_record);
    }
    return AQE_OK;
}

// Tile decomposition of one family window (kernels.hpp: one wave folds kTileOrdinals per tile).
void add_family(std::vector<DevFamily>& out, LaunchDesc& L, const aqe_family& f, uint64_t& out_pos, bool dense16) {
    if (f.ord_hi <= f.ord_lo) return;
    DevFamily d{};
    d.row0 = f.row0; d.pitch = f.pitch; d.seg_len = f.seg_len; d.step = f.step;
    d.ord_lo = f.ord_lo; d.ord_hi = f.ord_hi; d.group = f.group; d.flags = f.flags;
    uint64_t win_lo = f.ord_lo, win_hi = f.ord_hi;  // ordinals the tiles must cover
    uint64_t size_b = 0;
    if (f.flags & AQE_F_PAIR) {
        d.row0_b = f.row0_b; d.ord_lo_b = f.ord_lo_b; d.ord_hi_b = f.ord_hi_b;
        win_lo = std::min(win_lo, f.ord_lo_b);
        win_hi = std::max(win_hi, f.ord_hi_b);
        size_b = f.ord_hi_b - f.ord_lo_b;
    }
    const uint64_t tile = dense16 ? tile_ordinals(f.step, f.flags, f.seg_len) : kTileOrdinals;
    const uint64_t s_lo = win_lo / f.seg_len, s_hi = (win_hi - 1) / f.seg_len;
    uint64_t ntiles;
    d.seg_lo = s_lo;
    if (s_lo == s_hi) {
        d.tiles_per_seg = 0;
        d.j_lo = (win_lo % f.seg_len) / tile;
        ntiles = ((win_hi - 1) % f.seg_len) / tile + 1 - d.j_lo;
    } else {
        d.tiles_per_seg = (f.seg_len + tile - 1) / tile;
        d.j_lo = 0;
        ntiles = (s_hi - s_lo + 1) * d.tiles_per_seg;
    }
    d.tile_begin = L.ntiles;
    d.out_begin = out_pos;
    out_pos += f.ord_hi - f.ord_lo;
    d.out_begin_b = out_pos;
    out_pos += size_b;
    L.ntiles += ntiles;
    L.nfam += 1;
    L.samples += (f.ord_hi - f.ord_lo) + size_b;
    out.push_back(d);
}

FoldParams fold_params(const aqe_plan* p, bool topup) {
    FoldParams f{};
    f.shift = p->ctx->shift;
    f.z = p->host.clt.z;
    f.e = p->host.clt.e;
    f.base = p->host.clt.base;
    f.is_clt = p->host.is_clt ? 1 : 0;
    f.is_topup = topup ? 1 : 0;
    return f;
}

FinalizeParams finalize_params(const aqe_plan* p) {
    FinalizeParams f{};
    f.n_global = p->q.row_hi > p->q.row_lo ? p->q.row_hi - p->q.row_lo : p->ctx->n_global;  // a row window is the table
    f.pct = p->q.sample_percent;
    f.shift = p->ctx->shift;
    f.agg = p->q.agg;
    f.convention = p->q.convention;
    f.is_exact = p->q.method == AQE_M_EXACT;
    f.is_clt = p->host.is_clt;
    return f;
}

SweepCommon sweep_common(const aqe_plan* p, const DevFamily* fams, uint32_t nfam) {
    const aqe_ctx* c = p->ctx;
    SweepCommon s{};
    s.amount = p->host.on_sorted ? c->sorted_amount : c->amount;
    s.shard_lo = c->shard_lo;
    s.fams = fams;
    s.nfam = nfam;
    s.has_where = p->q.has_where ? 1 : 0;
    s.wmin = p->q.where_min;
    s.wmax = p->q.where_max;
    s.shift = c->shift;
    s.dense16 = c->dense16 ? 1 : 0;
    return s;
}

// `index` is the launch's position in the query: rounds 0..R-1, then the top-up.  The first launch
// folds into a zeroed state (no memset), later CLT launches test should_stop on entry, and in the fused
// single-GPU form the last launch also writes the result.
RoundLaunch round_launch(const aqe_plan* p, const LaunchDesc& L, uint32_t index, bool topup, bool fused, double* out_vec) {
    RoundLaunch a{};
    a.sw = sweep_common(p, p->d_fams ? p->d_fams + L.fam_offset : nullptr, L.nfam);
    a.ntiles = L.ntiles;
    a.partials = p->partials;
    a.counter = p->counter;
    a.out_vec = out_vec;
    a.state = p->d_state;
    a.fused = fused ? 1 : 0;
    a.check_stop = (p->host.is_clt && index > 0 && !topup) ? 1 : 0;
    a.reset_state = (index == 0 && !topup) ? 1 : 0;
    const uint32_t last = static_cast<uint32_t>(p->rounds.size()) - (p->host.has_topup ? 0u : 1u);
    a.do_finalize = (fused && index == last) ? 1 : 0;
    a.fold = fold_params(p, topup);
    a.fin = finalize_params(p);
    a.result = p->d_result;
    return a;
}

int plan_is_current(aqe_plan* p) {
    if (!p || !p->ctx) return AQE_ERR_INVALID;
    if (p->table_epoch != p->ctx->table_epoch)
        return fail(p->ctx, AQE_ERR_INVALID, "plan was created for a table that has since been replaced");
    return AQE_OK;
}

hipStream_t pick(aqe_plan* p, void* stream) { return stream ? static_cast<hipStream_t>(stream) : p->ctx->stream; }

int enqueue_launch(aqe_plan* p, const LaunchDesc& L, uint32_t index, bool topup, bool fused, double* out_vec, hipStream_t s) {
    aqe_ctx* c = p->ctx;
    if (!topup && index == 0) p->last_exec = 0;
    RoundLaunch a = round_launch(p, L, index, topup, fused, out_vec);
    const bool prof = p->profile && 2 * (p->lev_used + 1) <= p->lev.size();
    hipEvent_t e0 = prof ? p->lev[2 * p->lev_used] : nullptr, e1 = prof ? p->lev[2 * p->lev_used + 1] : nullptr;
    if (p->host.is_random && !topup) HIPCHK(c, launch_indexed(a, p->d_idx, p->host.random_idx.size(), s, e0, e1));
    else HIPCHK(c, launch_round(a, s, e0, e1));
    if (prof) p->lev_used++;
    return AQE_OK;
}

// Lay the plan's rounds (optionally the top-up as one more slot) out as ONE tile list and work out which
// workgroups own tiles of which slot.
int build_sweep_form(aqe_plan* p, bool with_topup_slot, SweepForm& F) {
    aqe_ctx* c = p->ctx;
    std::vector<const LaunchDesc*> slots;
    for (const auto& L : p->rounds) slots.push_back(&L);
    if (with_topup_slot && p->host.has_topup) slots.push_back(&p->topup);
    const size_t S = slots.size();
    uint64_t tiles = 0;
    for (size_t r = 0; r < S; ++r) {
        const LaunchDesc& L = *slots[r];
        F.round_begin[r] = tiles;
        for (uint32_t i = 0; i < L.nfam; ++i) {
            DevFamily d = p->h_fams[L.fam_offset + i];
            d.tile_begin += tiles;
            d.flags &= ~AQE_F_TOPUP;  // swept whole: the replay decides whether the top-up counts
            F.h_fams.push_back(d);
        }
        tiles += L.ntiles;
        F.samples += L.samples;
    }
    F.round_begin[S] = tiles;
    F.ntiles = tiles;
    F.slots = static_cast<uint32_t>(S);
    // Every wave but the monitor (wave 0 of workgroup 0) is a sweeper: sweeper v (physical wave v + 1) owns
    // tiles v, v + V, ...  The workgroups that own tiles of a slot form ONE cyclic run of workgroup ids (tiles
    // are consecutive, sweepers cyclic): find it by enumeration and insist on it — the monitor waits for
    // exactly these workgroups.
    const uint64_t G = c->persist_grid, V = G * kPersistWaves - 1;
    auto sweeper_has = [&](uint64_t v, uint64_t b0, uint64_t b1) { const uint64_t m0 = b0 % V; return b0 + (v >= m0 ? v - m0 : v + V - m0) < b1; };
    for (size_t r = 0; r < S; ++r) {
        F.round_mod[r] = static_cast<uint32_t>(F.round_begin[r] % V);
        std::vector<char> member(G, 0);
        uint64_t members = 0;
        for (uint64_t b = 0; b < G; ++b) {
            for (uint64_t j = 0; j < kPersistWaves && !member[b]; ++j) {
                const uint64_t phys = b * kPersistWaves + j;
                if (phys != 0 && sweeper_has(phys - 1, F.round_begin[r], F.round_begin[r + 1])) member[b] = 1;
            }
            members += member[b];
        }
        uint64_t first = 0;
        if (members != 0 && members != G) {
            uint64_t starts = 0;
            for (uint64_t b = 0; b < G; ++b)
                if (member[b] && !member[(b + G - 1) % G]) { first = b; ++starts; }
            if (starts != 1) return fail(c, AQE_ERR_INVALID, "internal: persistent-sweep participation is not one cyclic run");
        }
        for (uint64_t i = 0; i < members; ++i)
            if (!member[(first + i) % G]) return fail(c, AQE_ERR_INVALID, "internal: persistent-sweep participation is not one cyclic run");
        F.part_first[r] = static_cast<uint32_t>(first);
        F.part_count[r] = static_cast<uint32_t>(members);
        F.step_begin[r + 1] = F.step_begin[r] + static_cast<uint32_t>((members + 7) / 8);
    }
    F.round_mod[S] = static_cast<uint32_t>(F.round_begin[S] % V);
    if (!F.h_fams.empty()) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&F.d_fams), F.h_fams.size() * sizeof(DevFamily)));
        HIPCHK(c, hipMemcpy(F.d_fams, F.h_fams.data(), F.h_fams.size() * sizeof(DevFamily), hipMemcpyHostToDevice));
    }
    // the monitor reads whole windows of kDecSteps steps: keep one window of slack behind the last slot.
    // Pad slots (a round's run rounded up to 8) are never written: zero data, flag word "always published".
    std::vector<uint64_t> init(static_cast<size_t>(kVec) * 8 * (static_cast<size_t>(F.step_begin[S]) + kDecSteps), 0);
    for (size_t r = 0; r < S; ++r)
        for (size_t slot = 8 * static_cast<size_t>(F.step_begin[r]) + F.part_count[r]; slot < 8 * static_cast<size_t>(F.step_begin[r + 1]); ++slot)
            init[slot * kVec + 7] = kSlotAlways;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&F.d_ppart), init.size() * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpy(F.d_ppart, init.data(), init.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    F.ok = true;
    return AQE_OK;
}

int cached_plan(aqe_ctx* c, const aqe_query* q, aqe_plan** out);
int enqueue_all(aqe_plan* p, hipStream_t s, bool timed);

// GROUP BY needs the key column as SoA int32 on the device: from the resident 32-byte rows, or — for a table made
// by aqe_generate_synthetic — from the row number.  Built on first use, kept until the table changes.
int ensure_keys(aqe_ctx* c, int column) {
    const int k = column - 1;
    if (c->keycol[k] || c->n_local == 0) return AQE_OK;
    if (!c->aos && !c->synthetic)
        return fail(c, AQE_ERR_UNSUPPORTED, "grouped reduction needs the key columns: stage the table with AQE_STAGE_KEEP_AOS");
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->keycol[k]), c->n_local * sizeof(int32_t)));
    c->hbm_bytes += c->n_local * sizeof(int32_t);
    if (c->aos) HIPCHK(c, launch_extract_key(c->aos, c->keycol[k], c->n_local, column, c->stream));
    else HIPCHK(c, launch_synth_key(c->keycol[k], c->n_local, c->shard_lo, column, c->stream));
    int32_t* d_range = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_range), 2 * sizeof(int32_t)));
    int32_t init[2] = {std::numeric_limits<int32_t>::max(), std::numeric_limits<int32_t>::min()};
    hipError_t e = hipMemcpyAsync(d_range, init, sizeof init, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = launch_key_range(c->keycol[k], c->n_local, d_range, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(init, d_range, sizeof init, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_range);
    if (e != hipSuccess) return fail(c, AQE_ERR_HIP, std::string("key range: ") + hipGetErrorString(e));
    c->key_min[k] = init[0];
    c->key_max[k] = init[1];
    return AQE_OK;
}
int fetch(aqe_plan* p, aqe_result* out, hipStream_t s);

// adaptive_block_sample's pre-pass (DB.cpp:1291-1308): population variance of each of the ten zones from raw
// moments, var = Q/n - (S/n)^2 — ten exact window scans on the device, kept until the table changes.
int ensure_zone_variances(aqe_ctx* c) {
    if (c->zone_var_valid) return AQE_OK;
    const uint64_t zone_size = c->n_global / 10;
    if (zone_size == 0) return fail(c, AQE_ERR_INVALID, "adaptive_block_sample: needs at least 10 rows");
    for (uint64_t z = 0; z < 10; ++z) {
        aqe_query q;
        aqe_query_defaults(&q);
        q.method = AQE_M_EXACT;
        q.sample_percent = 100.0;
        q.row_lo = z * zone_size;
        q.row_hi = std::min(q.row_lo + zone_size, c->n_global);
        aqe_plan* p = nullptr;
        aqe_result r;
        int rc = cached_plan(c, &q, &p);
        if (rc == AQE_OK) rc = enqueue_all(p, c->stream, false);
        if (rc == AQE_OK) rc = fetch(p, &r, c->stream);
        if (rc != AQE_OK) return rc;
        const double cnt = static_cast<double>(q.row_hi - q.row_lo), mean = r.sum / cnt;
        c->zone_var[z] = (r.sumsq / cnt) - (mean * mean);
    }
    c->zone_var_valid = true;
    return AQE_OK;
}

// stratified_block_sample's pre-pass (DB.cpp:1342-1345): the amount column sorted ascending + its row permutation.
int ensure_sorted(aqe_ctx* c) {
    if (c->sorted_amount || c->n_local == 0) return AQE_OK;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->sorted_amount), (c->n_local + 1) * sizeof(double)));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->sorted_row), c->n_local * sizeof(uint32_t)));
    hipError_t e = sort_amounts(c->amount, c->n_local, c->sorted_amount, c->sorted_row, c->stream);
    if (e != hipSuccess) {
        (void)hipFree(c->sorted_amount); (void)hipFree(c->sorted_row);
        c->sorted_amount = nullptr; c->sorted_row = nullptr;
        return fail(c, AQE_ERR_HIP, std::string("sorting the amount column: ") + hipGetErrorString(e));
    }
    c->hbm_bytes += (c->n_local + 1) * sizeof(double) + c->n_local * sizeof(uint32_t);
    return AQE_OK;
}

int create_plan(aqe_ctx* c, const aqe_query* q, aqe_plan** out) {
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    if (q->agg < AQE_SUM || q->agg > AQE_COUNT) return fail(c, AQE_ERR_INVALID, "agg must be AQE_SUM, AQE_AVG or AQE_COUNT");
    if (q->convention < AQE_EST_CLI || q->convention > AQE_EST_RAW) return fail(c, AQE_ERR_INVALID, "convention must be AQE_EST_CLI, AQE_EST_CPP or AQE_EST_RAW");
    if (q->has_where && (q->where_min != q->where_min || q->where_max != q->where_max)) return fail(c, AQE_ERR_INVALID, "WHERE bound is NaN");
    std::unique_ptr<aqe_plan, void (*)(aqe_plan*)> p(new aqe_plan(), destroy_plan);
    p->ctx = c;
    p->q = *q;
    p->table_epoch = c->table_epoch;
    std::string err;
    const double* zone_var = nullptr;
    if (q->method == AQE_M_ADAPTIVE_BLOCK || q->method == AQE_M_STRATIFIED_BLOCK) {
        if (c->shard_lo != 0 || c->n_local != c->n_global)
            return fail(c, AQE_ERR_UNSUPPORTED, "adaptive/stratified block samplers need the whole table in this context (they need a global variance pass / sort)");
        int rc0 = q->method == AQE_M_ADAPTIVE_BLOCK ? ensure_zone_variances(c) : ensure_sorted(c);
        if (rc0 != AQE_OK) return rc0;
        zone_var = q->method == AQE_M_ADAPTIVE_BLOCK ? c->zone_var : nullptr;
    }
    int rc = build_plan(*q, c->n_global, ClipWindow{c->shard_lo, c->shard_lo + c->n_local}, p->host, err, zone_var);
    if (rc != AQE_OK) return fail(c, rc, err);
    uint64_t out_pos = 0;
    for (const auto& rf : p->host.round_fams) {
        LaunchDesc L;
        L.fam_offset = p->h_fams.size();
        for (const auto& f : rf) add_family(p->h_fams, L, f, out_pos, c->dense16);
        p->rounds.push_back(L);
    }
    if (p->host.is_random) {
        LaunchDesc L;
        L.samples = p->host.random_idx.size();
        p->rounds.assign(1, L);
    }
    if (p->host.has_topup) {
        p->topup.fam_offset = p->h_fams.size();
        for (const auto& f : p->host.topup_fams) add_family(p->h_fams, p->topup, f, out_pos, c->dense16);
    }
    if (!p->h_fams.empty()) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_fams), p->h_fams.size() * sizeof(DevFamily)));
        HIPCHK(c, hipMemcpy(p->d_fams, p->h_fams.data(), p->h_fams.size() * sizeof(DevFamily), hipMemcpyHostToDevice));
    }
    if (!p->host.random_idx.empty()) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_idx), p->host.random_idx.size() * sizeof(uint64_t)));
        HIPCHK(c, hipMemcpy(p->d_idx, p->host.random_idx.data(), p->host.random_idx.size() * sizeof(uint64_t),
                            hipMemcpyHostToDevice));
    }
    {   // persistent single-launch forms of the rounds
        const size_t R = p->rounds.size();
        const bool multi = !p->host.is_random && R >= 2 && c->persist_grid > 0;
        const bool whole = c->shard_lo == 0 && c->n_local == c->n_global;
        bool every_round_has_tiles = true;
        for (size_t r = 0; r < R; ++r) every_round_has_tiles = every_round_has_tiles && p->rounds[r].ntiles > 0;
        if (multi && whole && every_round_has_tiles && R <= static_cast<size_t>(kMaxPersistRounds) && !(q->flags & AQE_Q_NO_PERSIST)) {
            int rc2 = build_sweep_form(p.get(), false, p->decide);
            if (rc2 != AQE_OK) return rc2;
            p->persist = true;
        }
        if (multi && R <= static_cast<size_t>(kMaxPersistRounds)) {
            int rc2 = build_sweep_form(p.get(), false, p->totals);
            if (rc2 != AQE_OK) return rc2;
        }
        if (p->decide.ok || p->totals.ok) {
            HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_ctl), sizeof(PersistCtl)));
            HIPCHK(c, hipMemset(p->d_ctl, 0, sizeof(PersistCtl)));
            HIPCHK(c, hipMalloc(&p->d_rehearsal, sizeof(QueryState) + sizeof(aqe_result)));
        }
    }
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->partials), sizeof(double) * kVec * kMaxBlocks));
    HIPCHK(c, hipMemset(p->partials, 0, sizeof(double) * kVec * kMaxBlocks));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->counter), sizeof(unsigned) * kCounterWords));
    HIPCHK(c, hipMemset(p->counter, 0, sizeof(unsigned) * kCounterWords));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_state), sizeof(QueryState)));
    HIPCHK(c, hipMemset(p->d_state, 0, sizeof(QueryState)));
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&p->h_result), sizeof(aqe_result), hipHostMallocMapped));
    std::memset(p->h_result, 0, sizeof(aqe_result));
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_result), p->h_result, 0));
    HIPCHK(c, hipEventCreate(&p->ev0));
    HIPCHK(c, hipEventCreate(&p->ev1));
    *out = p.release();
    return AQE_OK;
}

int cached_plan(aqe_ctx* c, const aqe_query* q, aqe_plan** out) {
    for (auto& kv : c->cache)
        if (std::memcmp(&kv.first, q, sizeof(aqe_query)) == 0) { *out = kv.second; return AQE_OK; }
    aqe_plan* p = nullptr;
    int rc = create_plan(c, q, &p);
    if (rc != AQE_OK) return rc;
    if (c->cache.size() >= 64) { destroy_plan(c->cache.front().second); c->cache.erase(c->cache.begin()); }
    c->cache.emplace_back(*q, p);
    *out = p;
    return AQE_OK;
}

int launch_form(aqe_plan* p, const SweepForm& F, bool totals_only, double* out_totals, hipStream_t s) {
    aqe_ctx* c = p->ctx;
    PersistLaunch a{};
    a.sw = sweep_common(p, F.d_fams, static_cast<uint32_t>(F.h_fams.size()));
    a.ntiles = F.ntiles;
    for (uint32_t r = 0; r <= F.slots; ++r) a.round_begin[r] = F.round_begin[r];
    for (uint32_t r = 0; r < F.slots; ++r) { a.part_first[r] = F.part_first[r]; a.part_count[r] = F.part_count[r]; }
    for (uint32_t r = 0; r <= F.slots; ++r) { a.step_begin[r] = F.step_begin[r]; a.round_mod[r] = F.round_mod[r]; }
    a.rounds = F.slots;
    a.epoch = c->epoch++;
    a.ctl = p->d_ctl;
    a.partials = F.d_ppart;
    a.state = p->d_state;
    a.fold = fold_params(p, false);
    a.fin = finalize_params(p);
    a.result = p->d_result;
    a.rehearsal_state = static_cast<QueryState*>(p->d_rehearsal);
    a.rehearsal_result = reinterpret_cast<aqe_result*>(static_cast<char*>(p->d_rehearsal) + sizeof(QueryState));
    a.stamps = c->d_stamps;
    a.finalize_here = p->host.has_topup ? 0u : 1u;
    a.totals_only = totals_only ? 1u : 0u;
    p->last_exec = totals_only ? 2 : 1;
    a.out_totals = out_totals;
    a.inline_fams = F.h_fams.size() <= static_cast<size_t>(kPersistInlineFams) ? 1u : 0u;
    if (a.inline_fams) std::copy(F.h_fams.begin(), F.h_fams.end(), a.fams);
    if (c->d_stamps) {
        HIPCHK(c, hipMemsetAsync(c->d_stamps, 0, 8 * (8 * static_cast<size_t>(c->persist_grid) * kPersistWaves + 8 * kMaxPersistRounds), s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    const bool prof = p->profile && 2 * (p->lev_used + 1) <= p->lev.size();
    HIPCHK(c, launch_sweep_persist(a, c->persist_grid, s, prof ? p->lev[2 * p->lev_used] : nullptr, prof ? p->lev[2 * p->lev_used + 1] : nullptr));
    if (prof) p->lev_used++;
    return AQE_OK;
}

int enqueue_all(aqe_plan* p, hipStream_t s, bool timed) {
    aqe_ctx* c = p->ctx;
    if (timed) HIPCHK(c, hipEventRecord(p->ev0, s));  // an event record is a queue packet: off the throughput path
    p->lev_used = 0;
    bool topup_done = false;
    if (p->rounds.empty()) {  // nothing to sample (empty table / zero target): a zero state, finalized
        HIPCHK(c, hipMemsetAsync(p->d_state, 0, sizeof(QueryState), s));
        HIPCHK(c, launch_finalize(p->d_state, finalize_params(p), p->d_result, s));
    } else {
        if (p->persist) {
            int rc = launch_form(p, p->decide, false, nullptr, s);
            if (rc != AQE_OK) return rc;
        } else if (p->rounds.size() >= kGraphMinRounds && p->rounds.size() <= kGraphMaxRounds && !p->profile && !std::getenv("AQE_NO_GRAPH")) {
            // One launch per round is a launch-bound loop (every launch after the stop is a device-side no-op): it is
            // captured ONCE per plan into a HIP graph — the launches' arguments never change — and replayed.
            if (!p->round_graph) {
                hipGraph_t g = nullptr;
                HIPCHK(c, hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                int rc = AQE_OK;
                for (uint32_t i = 0; i < p->rounds.size() && rc == AQE_OK; ++i) rc = enqueue_launch(p, p->rounds[i], i, false, true, nullptr, s);
                if (rc == AQE_OK && p->host.has_topup)
                    rc = enqueue_launch(p, p->topup, static_cast<uint32_t>(p->rounds.size()), true, true, nullptr, s);
                hipError_t e = hipStreamEndCapture(s, &g);  // always ends the capture, also after a failed launch
                if (rc != AQE_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
                if (e == hipSuccess) e = hipGraphInstantiate(&p->round_graph, g, nullptr, nullptr, 0);
                if (g) (void)hipGraphDestroy(g);
                if (e != hipSuccess) { p->round_graph = nullptr; return fail(c, AQE_ERR_HIP, std::string("capturing the round launches: ") + hipGetErrorString(e)); }
            }
            p->last_exec = 0;
            HIPCHK(c, hipGraphLaunch(p->round_graph, s));
            topup_done = true;
        } else {
            for (uint32_t i = 0; i < p->rounds.size(); ++i) {
                int rc = enqueue_launch(p, p->rounds[i], i, false, true, nullptr, s);
                if (rc != AQE_OK) return rc;
            }
        }
        if (p->host.has_topup && !topup_done) {
            int rc = enqueue_launch(p, p->topup, static_cast<uint32_t>(p->rounds.size()), true, true, nullptr, s);
            if (rc != AQE_OK) return rc;
        }
    }
    if (timed) HIPCHK(c, hipEventRecord(p->ev1, s));
    p->timed = timed;
    return AQE_OK;
}

int fetch(aqe_plan* p, aqe_result* out, hipStream_t s) {
    aqe_ctx* c = p->ctx;
    HIPCHK(c, hipStreamSynchronize(s));
    *out = *p->h_result;
    if (c->d_stamps && p->persist) {
        const size_t W = static_cast<size_t>(c->persist_grid) * kPersistWaves;
        std::vector<unsigned long long> st(8 * W + 8 * kMaxPersistRounds);
        (void)hipMemcpy(st.data(), c->d_stamps, st.size() * 8, hipMemcpyDeviceToHost);
        if (FILE* f = std::fopen(std::getenv("AQE_PERSIST_STAMPS"), "a")) {
            unsigned long long t0 = ~0ull, s_hi = 0, f_lo = ~0ull, f_hi = 0, l_hi = 0, e_hi = 0, h_hi = 0, p_hi = 0, d_hi = 0;
            for (size_t w = 0; w < W; ++w) {
                const unsigned long long* q = &st[8 * w];
                if (q[0]) { t0 = std::min(t0, q[0]); s_hi = std::max(s_hi, q[0]); }
                if (q[1]) { f_lo = std::min(f_lo, q[1]); f_hi = std::max(f_hi, q[1]); }
                l_hi = std::max(l_hi, q[2]);
                e_hi = std::max(e_hi, q[3]);
                h_hi = std::max(h_hi, q[4]);
                p_hi = std::max(p_hi, q[5]);
                d_hi = std::max(d_hi, q[6]);
            }
            auto us = [&](unsigned long long v) { return v == 0 || v == ~0ull ? -1.0 : (static_cast<double>(v) - static_cast<double>(t0)) / 100.0; };
            std::fprintf(f, "starts ..%.2f first-tile %.2f..%.2f last-tile %.2f handed %.2f stored %.2f drained %.2f end %.2f |", us(s_hi), us(f_lo),
                         us(f_hi), us(l_hi), us(h_hi), us(p_hi), us(d_hi), us(e_hi));
            for (size_t r = 0; r < p->rounds.size(); ++r) {
                const unsigned long long* q = &st[8 * W + 8 * r];
                if (r == 0 && q[6]) std::fprintf(f, " rehearsal done %.2f |", us(q[6]));
                if (q[3]) std::fprintf(f, " ..r%zu: seen %.2f folded %.2f judged %.2f |", r, us(q[3]), us(q[4]), us(q[5]));
            }
            std::fprintf(f, "\n");
            std::fclose(f);
        }
    }
    if (out->device_status != 0) {  // the monitor gave up waiting for a workgroup's partial
        return fail(c, AQE_ERR_HIP, "device-side round protocol timed out");
    }
    if (p->timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->ev0, p->ev1) == hipSuccess) out->kernel_ms = ms;
    }
    return AQE_OK;
}

// Worker threads that fill the pinned bounce buffers: one host core reads rows at ~13 GB/s, a fifth of what the
// PCIe link takes, so the fill of every chunk is split over the host's cores (the loader).
class FillPool {
  public:
    explicit FillPool(unsigned n) {
        for (unsigned i = 0; i < n; ++i)
            workers_.emplace_back([this, i, n] {
                unsigned seen = 0;
                for (;;) {
                    std::unique_lock<std::mutex> lk(m_);
                    wake_.wait(lk, [&] { return stop_ || gen_ != seen; });
                    if (stop_) return;
                    seen = gen_;
                    lk.unlock();
                    job_(i, n);
                    lk.lock();
                    if (--pending_ == 0) done_.notify_one();
                }
            });
    }
    ~FillPool() {
        { std::lock_guard<std::mutex> lk(m_); stop_ = true; }
        wake_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // runs f(part, parts) on every worker and returns when all are done
    void run(std::function<void(unsigned, unsigned)> f) {
        std::unique_lock<std::mutex> lk(m_);
        job_ = std::move(f);
        pending_ = static_cast<unsigned>(workers_.size());
        ++gen_;
        wake_.notify_all();
        done_.wait(lk, [&] { return pending_ == 0; });
    }

  private:
    std::vector<std::thread> workers_;
    std::mutex m_;
    std::condition_variable wake_, done_;
    std::function<void(unsigned, unsigned)> job_;
    unsigned gen_ = 0, pending_ = 0;
    bool stop_ = false;
};

int stage_from_host(aqe_ctx* c, const aqe_record* rows, uint64_t n_local, uint64_t shard_lo, uint64_t n_global,
                    uint32_t flags) {
    if (shard_lo + n_local > n_global) return fail(c, AQE_ERR_INVALID, "shard exceeds the table");
    if (n_local && !rows) return fail(c, AQE_ERR_INVALID, "null rows");
    const bool keep = flags & AQE_STAGE_KEEP_AOS;
    int rc = alloc_table(c, n_local, keep);
    if (rc != AQE_OK) return rc;
    c->n_global = n_global;
    c->shard_lo = shard_lo;
    c->n_local = n_local;
    c->staged = true;
    c->shift = shift_of_rows(rows, n_local);  // shards other than the first are given the table's value (aqe_set_shift)
    if (n_local == 0) return AQE_OK;
    // Double-buffered pinned bounce: the CPU fills buffer b while the DMA engine drains buffer b^1.
    // Without KEEP_AOS only the amount column crosses PCIe (8 of every 32 bytes).
    const size_t row_bytes = keep ? sizeof(aqe_record) : sizeof(double);
    struct Bounce {  // two pinned buffers + their "drained" events, released on every exit path
        void* pinned[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        ~Bounce() {
            for (int b = 0; b < 2; ++b) {
                if (pinned[b]) (void)hipHostFree(pinned[b]);
                if (done[b]) (void)hipEventDestroy(done[b]);
            }
        }
    } bounce;
    void** pinned = bounce.pinned;
    hipEvent_t* done = bounce.done;
    for (int b = 0; b < 2; ++b) {
        if (hipHostMalloc(&pinned[b], std::min<uint64_t>(kStageChunkRows, n_local) * row_bytes, hipHostMallocDefault) != hipSuccess ||
            hipEventCreateWithFlags(&done[b], hipEventDisableTiming) != hipSuccess) {
            free_table(c);
            return fail(c, AQE_ERR_HIP, "staging: cannot allocate pinned bounce buffers");
        }
    }
    int status = AQE_OK;
    std::atomic<bool> dense_ids{true};
    const int64_t id0 = rows[0].id;
    const unsigned hw = std::thread::hardware_concurrency();
    FillPool pool(n_local < (1u << 18) ? 1u : std::max(1u, std::min(16u, hw ? hw : 4u)));
    for (uint64_t off = 0, k = 0; off < n_local && status == AQE_OK; off += kStageChunkRows, ++k) {
        const int b = static_cast<int>(k & 1);
        const uint64_t m = std::min<uint64_t>(kStageChunkRows, n_local - off);
        if (k >= 2 && hipEventSynchronize(done[b]) != hipSuccess) { status = fail(c, AQE_ERR_HIP, "event sync"); break; }
        hipError_t e;
        void* const dst_buf = pinned[b];
        pool.run([&, dst_buf](unsigned part, unsigned parts) {  // rows [lo, hi) of the chunk: fill + dense-id check
            const uint64_t lo = m * part / parts, hi = m * (part + 1) / parts;
            bool dense = true;
            for (uint64_t i = lo; i < hi; ++i) dense = dense && rows[off + i].id == id0 + static_cast<int64_t>(off + i);
            if (!dense) dense_ids.store(false, std::memory_order_relaxed);
            if (keep) {
                std::memcpy(static_cast<aqe_record*>(dst_buf) + lo, rows + off + lo, (hi - lo) * sizeof(aqe_record));
            } else {
                double* dst = static_cast<double*>(dst_buf);
                for (uint64_t i = lo; i < hi; ++i) dst[i] = rows[off + i].amount;
            }
        });
        if (keep) {
            e = hipMemcpyAsync(c->aos + off, pinned[b], m * sizeof(aqe_record), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = launch_split_amount(c->aos + off, c->amount + off, m, c->stream);
        } else {
            e = hipMemcpyAsync(c->amount + off, pinned[b], m * sizeof(double), hipMemcpyHostToDevice, c->stream);
        }
        if (e == hipSuccess) e = hipEventRecord(done[b], c->stream);
        if (e != hipSuccess) status = fail(c, AQE_ERR_HIP, std::string("staging: ") + hipGetErrorString(e));
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess && status == AQE_OK) status = fail(c, AQE_ERR_HIP, std::string("staging sync: ") + hipGetErrorString(e));
    if (status != AQE_OK) { free_table(c); return status; }
    c->ids_dense = dense_ids.load();
    c->first_id = id0 - static_cast<int64_t>(shard_lo);  // id of global row 0 when the ids are dense
    return status;
}

struct MappedFile {
    void* base = MAP_FAILED;
    size_t bytes = 0;
    int fd = -1;
    ~MappedFile() {
        if (base != MAP_FAILED) munmap(base, bytes);
        if (fd >= 0) close(fd);
    }
};

int open_db_file(aqe_ctx* c, const char* path, MappedFile& mf, uint64_t& count) {
    mf.fd = open(path, O_RDONLY);
    if (mf.fd < 0) return fail(c, AQE_ERR_IO, std::string("cannot open ") + path);
    struct stat st;
    if (fstat(mf.fd, &st) != 0 || st.st_size < 24) return fail(c, AQE_ERR_IO, std::string("not an aqe database file: ") + path);
    mf.bytes = static_cast<size_t>(st.st_size);
    mf.base = mmap(nullptr, mf.bytes, PROT_READ, MAP_PRIVATE, mf.fd, 0);
    if (mf.base == MAP_FAILED) return fail(c, AQE_ERR_IO, std::string("mmap failed: ") + path);
    uint64_t hdr[3];  // size_t total | size_t height | size_t count, DB.cpp:669-676
    std::memcpy(hdr, mf.base, sizeof hdr);
    count = hdr[2];
    if (24 + count * sizeof(aqe_record) > mf.bytes) return fail(c, AQE_ERR_IO, std::string("truncated database file: ") + path);
    return AQE_OK;
}

}  // namespace

extern "C" {

int aqe_abi_version(void) { return AQE_ABI_VERSION; }

const char* aqe_status_string(int s) {
    switch (s) {
        case AQE_OK: return "ok";
        case AQE_ERR_INVALID: return "invalid argument";
        case AQE_ERR_HIP: return "HIP error";
        case AQE_ERR_NO_DEVICE: return "no usable gfx950 device";
        case AQE_ERR_NO_TABLE: return "no table staged";
        case AQE_ERR_IO: return "I/O error";
        case AQE_ERR_CAPACITY: return "output buffer too small";
        case AQE_ERR_UNSUPPORTED: return "unsupported";
        default: return "unknown status";
    }
}

const char* aqe_last_error(const aqe_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int aqe_create(int device_id, aqe_ctx** out) {
    if (!out) return fail(nullptr, AQE_ERR_INVALID, "out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(nullptr, AQE_ERR_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "count is 0"));
    if (device_id < 0 || device_id >= n) return fail(nullptr, AQE_ERR_INVALID, "device_id out of range");
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess) return fail(nullptr, AQE_ERR_NO_DEVICE, "cannot query device");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, AQE_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    std::unique_ptr<aqe_ctx> c(new aqe_ctx());
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) return fail(nullptr, AQE_ERR_HIP, "hipSetDevice failed");
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) return fail(nullptr, AQE_ERR_HIP, "stream creation failed");
    // one 16-wave workgroup per CU, rounded down to a power of two (the wave->tile map uses masks)
    c->persist_grid = 16;
    while (c->persist_grid * 2 <= static_cast<unsigned>(prop.multiProcessorCount) && c->persist_grid * 2 <= kMaxPersistGrid) c->persist_grid *= 2;
    if (std::getenv("AQE_PERSIST_STAMPS") &&
        hipMalloc(reinterpret_cast<void**>(&c->d_stamps), 8 * (8 * static_cast<size_t>(c->persist_grid) * kPersistWaves + 8 * kMaxPersistRounds)) != hipSuccess)
        return fail(nullptr, AQE_ERR_HIP, "stamp buffer allocation failed");
    *out = c.release();
    return AQE_OK;
}

void aqe_destroy(aqe_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    drop_cache(c);
    free_table(c);
    if (c->d_stamps) (void)hipFree(c->d_stamps);
    if (c->grp_partial) (void)hipFree(c->grp_partial);
    if (c->grp_out) (void)hipFree(c->grp_out);
    if (c->grp_bins) (void)hipFree(c->grp_bins);
    for (hipStream_t s : c->lanes) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int aqe_stage_records(aqe_ctx* c, const void* aos32, uint64_t n_local, uint64_t shard_lo, uint64_t n_global, uint32_t flags) {
    if (!c) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    return stage_from_host(c, static_cast<const aqe_record*>(aos32), n_local, shard_lo, n_global, flags);
}

int aqe_file_rows(const char* path, uint64_t* n_rows) {
    if (!path || !n_rows) return AQE_ERR_INVALID;
    FILE* f = std::fopen(path, "rb");
    if (!f) return fail(nullptr, AQE_ERR_IO, std::string("cannot open ") + path);
    uint64_t hdr[3];
    size_t got = std::fread(hdr, sizeof hdr, 1, f);
    std::fclose(f);
    if (got != 1) return fail(nullptr, AQE_ERR_IO, std::string("not an aqe database file: ") + path);
    *n_rows = hdr[2];
    return AQE_OK;
}

int aqe_stage_file(aqe_ctx* c, const char* path, uint64_t shard_lo, uint64_t n_local, uint32_t flags) {
    if (!c || !path) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    MappedFile mf;
    uint64_t count = 0;
    int rc = open_db_file(c, path, mf, count);
    if (rc != AQE_OK) return rc;
    if (shard_lo > count) return fail(c, AQE_ERR_INVALID, "shard_lo beyond the end of the file");
    if (n_local == 0) n_local = count - shard_lo;
    if (shard_lo + n_local > count) return fail(c, AQE_ERR_INVALID, "shard exceeds the file");
    (void)madvise(mf.base, mf.bytes, MADV_SEQUENTIAL);
    const aqe_record* rows = reinterpret_cast<const aqe_record*>(static_cast<const char*>(mf.base) + 24);
    rc = stage_from_host(c, rows + shard_lo, n_local, shard_lo, count, flags);
    if (rc == AQE_OK && count) c->shift = shift_of_rows(rows, count);  // from the table's head: identical on every shard
    return rc;
}

int aqe_save_file(aqe_ctx* c, const char* path) {
    if (!c || !path) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_local && !c->aos) return fail(c, AQE_ERR_UNSUPPORTED, "save needs the rows resident (AQE_STAGE_KEEP_AOS)");
    if (c->shard_lo != 0 || c->n_local != c->n_global) return fail(c, AQE_ERR_UNSUPPORTED, "save needs the whole table in this context");
    FILE* f = std::fopen(path, "wb");
    if (!f) return fail(c, AQE_ERR_IO, std::string("cannot create ") + path);
    uint64_t height = 1;  // informational: the reference rebuilds its tree on load (DB.cpp:703-710)
    for (uint64_t cap = 254; c->n_global > cap; cap *= 128) ++height;
    uint64_t hdr[3] = {c->n_global, height, c->n_global};
    bool ok = std::fwrite(hdr, sizeof hdr, 1, f) == 1;
    std::vector<aqe_record> buf(std::min<uint64_t>(kStageChunkRows, std::max<uint64_t>(c->n_local, 1)));
    for (uint64_t off = 0; ok && off < c->n_local; off += buf.size()) {
        uint64_t m = std::min<uint64_t>(buf.size(), c->n_local - off);
        if (hipMemcpy(buf.data(), c->aos + off, m * sizeof(aqe_record), hipMemcpyDeviceToHost) != hipSuccess) { ok = false; break; }
        ok = std::fwrite(buf.data(), sizeof(aqe_record), m, f) == m;
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? AQE_OK : fail(c, AQE_ERR_IO, std::string("write failed: ") + path);
}

int aqe_generate_synthetic(aqe_ctx* c, uint64_t n_local, uint64_t shard_lo, uint64_t n_global, uint64_t seed, uint32_t flags) {
    if (!c) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (shard_lo + n_local > n_global) return fail(c, AQE_ERR_INVALID, "shard exceeds the table");
    int rc = alloc_table(c, n_local, flags & AQE_STAGE_KEEP_AOS);
    if (rc != AQE_OK) return rc;
    c->n_global = n_global;
    c->shard_lo = shard_lo;
    c->n_local = n_local;
    c->staged = true;
    {   // shift from the table's first rows, evaluated with the expression the kernel uses (every shard agrees)
        const uint64_t m = std::min<uint64_t>(n_global, kShiftRows);
        double acc = 0.0;
        for (uint64_t i = 0; i < m; ++i) {
            uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
            z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
            z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
            z ^= z >> 31;
            acc += 1.0 + 999.0 * (static_cast<double>(z >> 11) * (1.0 / 9007199254740992.0));
        }
        c->shift = m ? acc / static_cast<double>(m) : 0.0;
    }
    c->ids_dense = true;  // id = row + 1
    c->first_id = 1;
    c->synthetic = true;
    HIPCHK(c, launch_synth(c->aos, c->amount, n_local, shard_lo, seed, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return AQE_OK;
}

int aqe_attach_device(aqe_ctx* c, const double* dev_amount, const void* dev_aos32, uint64_t n_local, uint64_t shard_lo,
                      uint64_t n_global, double shift) {
    if (!c) return AQE_ERR_INVALID;
    if (n_local && !dev_amount) return fail(c, AQE_ERR_INVALID, "null amount column");
    if (shard_lo + n_local > n_global) return fail(c, AQE_ERR_INVALID, "shard exceeds the table");
    free_table(c);
    drop_cache(c);
    c->owns_table = false;
    c->dense16 = n_local >= 2;  // caller-owned memory has no spare row
    c->staged = true;
    c->amount = const_cast<double*>(dev_amount);
    c->aos = static_cast<aqe_record*>(const_cast<void*>(dev_aos32));
    c->n_local = n_local;
    c->shard_lo = shard_lo;
    c->n_global = n_global;
    c->shift = shift;
    return AQE_OK;
}

int aqe_set_shift(aqe_ctx* c, double shift) {
    if (!c) return AQE_ERR_INVALID;
    c->shift = shift;
    drop_cache(c);
    return AQE_OK;
}

int aqe_table_info_get(const aqe_ctx* c, aqe_table_info* out) {
    if (!c || !out) return AQE_ERR_INVALID;
    out->global_rows = c->n_global;
    out->shard_lo = c->shard_lo;
    out->local_rows = c->n_local;
    out->shift = c->shift;
    out->has_aos = c->aos != nullptr;
    out->device_id = c->device;
    out->hbm_bytes = c->hbm_bytes;
    return AQE_OK;
}

int aqe_key_range_rows(aqe_ctx* c, int64_t id_min, int64_t id_max, uint64_t* row_lo, uint64_t* row_hi) {
    if (!c || !row_lo || !row_hi) return AQE_ERR_INVALID;
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    HIPCHK(c, hipSetDevice(c->device));
    *row_lo = *row_hi = 0;
    if (id_max < id_min || c->n_global == 0) return AQE_OK;
    if (c->ids_dense) {  // id = first_id + row
        const int64_t last = c->first_id + static_cast<int64_t>(c->n_global) - 1;
        if (id_max < c->first_id || id_min > last) return AQE_OK;
        *row_lo = static_cast<uint64_t>(std::max(id_min, c->first_id) - c->first_id);
        *row_hi = static_cast<uint64_t>(std::min(id_max, last) - c->first_id) + 1;
        return AQE_OK;
    }
    if (c->shard_lo != 0 || c->n_local != c->n_global) return fail(c, AQE_ERR_UNSUPPORTED, "key bounds on a shard need dense ids");
    if (!c->aos) return fail(c, AQE_ERR_UNSUPPORTED, "key bounds on non-dense ids need the rows resident (AQE_STAGE_KEEP_AOS)");
    uint64_t* d_out = nullptr;
    uint64_t h_out[2] = {0, 0};
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_out), sizeof h_out));
    hipError_t e = launch_id_bounds(c->aos, c->n_local, id_min, id_max, d_out, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(h_out, d_out, sizeof h_out, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, AQE_ERR_HIP, std::string("key bounds: ") + hipGetErrorString(e));
    *row_lo = h_out[0];
    *row_hi = std::max(h_out[0], h_out[1]);
    return AQE_OK;
}

int aqe_release_table(aqe_ctx* c) {
    if (!c) return AQE_ERR_INVALID;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    drop_cache(c);
    free_table(c);
    return AQE_OK;
}

// ---- host planning ------------------------------------------------------------------------------
void aqe_query_defaults(aqe_query* q) {
    if (!q) return;
    std::memset(q, 0, sizeof *q);
    q->method = AQE_M_MEMORY_STRIDE;
    q->agg = AQE_SUM;
    q->convention = AQE_EST_CLI;
    q->num_threads = 4;        // BIND:62-101 defaults
    q->sample_percent = 10.0;
    q->block_size = 1000;
    q->seed = 42;
    q->step_size = 2;
    q->check_interval = 10;
    q->confidence_level = 0.95;
    q->max_error_percent = 2.0;
    q->clt_growth = 1;
    q->block_size_max = 2000;  // adaptive_block_sample's max_block_size (BIND:79-80); its min is 500
}

int aqe_plan_families(const aqe_query* q, uint64_t n_global, uint64_t shard_lo, uint64_t shard_hi, uint32_t round,
                      aqe_family* fams, uint32_t cap, uint32_t* n_out, uint32_t* rounds_out, uint64_t* samples_out) {
    if (!q) return AQE_ERR_INVALID;
    HostPlan P;
    std::string err;
    int rc = build_plan(*q, n_global, ClipWindow{shard_lo, shard_hi}, P, err);
    if (rc != AQE_OK) return fail(nullptr, rc, err);
    if (rounds_out) *rounds_out = P.rounds;
    if (samples_out) *samples_out = P.global_samples;
    const std::vector<aqe_family>* src = nullptr;
    static const std::vector<aqe_family> none;
    if (P.is_random) src = &none;
    else if (round < P.round_fams.size()) src = &P.round_fams[round];
    else if (round == P.rounds && P.has_topup) src = &P.topup_fams;
    else src = &none;
    if (n_out) *n_out = static_cast<uint32_t>(src->size());
    if (fams) {
        if (cap < src->size()) return fail(nullptr, AQE_ERR_CAPACITY, "family buffer too small");
        std::copy(src->begin(), src->end(), fams);
    }
    return AQE_OK;
}

int aqe_plan_adaptive_families(const aqe_query* q, uint64_t n_global, const double* zone_var10, aqe_family* fams,
                               uint32_t cap, uint32_t* n_out, uint64_t* samples_out) {
    if (!q || !zone_var10) return AQE_ERR_INVALID;
    if (q->method != AQE_M_ADAPTIVE_BLOCK) return fail(nullptr, AQE_ERR_INVALID, "not an adaptive_block_sample query");
    HostPlan P;
    std::string err;
    int rc = build_plan(*q, n_global, ClipWindow{0, n_global}, P, err, zone_var10);
    if (rc != AQE_OK) return fail(nullptr, rc, err);
    const auto& src = P.round_fams.at(0);
    if (n_out) *n_out = static_cast<uint32_t>(src.size());
    if (samples_out) *samples_out = P.global_samples;
    if (fams) {
        if (cap < src.size()) return fail(nullptr, AQE_ERR_CAPACITY, "family buffer too small");
        std::copy(src.begin(), src.end(), fams);
    }
    return AQE_OK;
}

int aqe_plan_random_indices(uint64_t n_global, double pct, uint32_t seed, uint64_t shard_lo, uint64_t shard_hi,
                            uint64_t* out, uint64_t cap, uint64_t* n_out) {
    std::vector<uint64_t> idx;
    std::string err;
    int rc = random_pointer_indices(n_global, pct, seed, ClipWindow{shard_lo, shard_hi}, idx, err);
    if (rc != AQE_OK) return fail(nullptr, rc, err);
    if (n_out) *n_out = idx.size();
    if (out) {
        if (cap < idx.size()) return fail(nullptr, AQE_ERR_CAPACITY, "index buffer too small");
        std::copy(idx.begin(), idx.end(), out);
    }
    return AQE_OK;
}

int aqe_parse_where(const char* query, double* lo, double* hi) {
    double a, b;
    bool found = parse_where(query, &a, &b);
    if (lo) *lo = a;
    if (hi) *hi = b;
    return found ? 1 : 0;
}

double aqe_confidence_heuristic(double pct, uint64_t total) { return confidence_heuristic(pct, total); }
double aqe_error_to_sample_percent(double e) { return error_to_sample_percent(e); }

// ---- plans --------------------------------------------------------------------------------------
int aqe_plan_create(aqe_ctx* c, const aqe_query* q, aqe_plan** out) {
    if (!c || !q || !out) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    return create_plan(c, q, out);
}

void aqe_plan_destroy(aqe_plan* p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    destroy_plan(p);
}

int aqe_plan_rounds(const aqe_plan* p, uint32_t* rounds, int32_t* has_topup) {
    if (!p) return AQE_ERR_INVALID;
    if (rounds) *rounds = static_cast<uint32_t>(p->rounds.size());
    if (has_topup) *has_topup = p->host.has_topup ? 1 : 0;
    return AQE_OK;
}

int aqe_plan_reset(aqe_plan* p, void* stream) {
    int rc = plan_is_current(p);
    if (rc != AQE_OK) return rc;
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    hipStream_t s = pick(p, stream);
    if (p->rounds.empty()) HIPCHK(p->ctx, hipMemsetAsync(p->d_state, 0, sizeof(QueryState), s));
    p->timed = false;
    p->lev_used = 0;
    return AQE_OK;
}

int aqe_plan_enqueue_round(aqe_plan* p, uint32_t round, double* dev_vec, void* stream) {
    int rc = plan_is_current(p);
    if (rc != AQE_OK) return rc;
    if (!dev_vec) return fail(p->ctx, AQE_ERR_INVALID, "dev_vec is null");
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    const bool topup = round == p->rounds.size() && p->host.has_topup;
    if (!topup && round >= p->rounds.size()) return fail(p->ctx, AQE_ERR_INVALID, "round out of range");
    return enqueue_launch(p, topup ? p->topup : p->rounds[round], round, topup, false, dev_vec, pick(p, stream));
}

int aqe_plan_enqueue_update(aqe_plan* p, uint32_t round, const double* dev_vec, void* stream) {
    int rc = plan_is_current(p);
    if (rc != AQE_OK) return rc;
    if (!dev_vec) return fail(p->ctx, AQE_ERR_INVALID, "dev_vec is null");
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    const bool topup = round == p->rounds.size() && p->host.has_topup;
    if (!topup && round >= p->rounds.size()) return fail(p->ctx, AQE_ERR_INVALID, "round out of range");
    HIPCHK(p->ctx, launch_update(p->d_state, dev_vec, fold_params(p, topup), (round == 0 && !topup) ? 1 : 0, pick(p, stream)));
    return AQE_OK;
}

int aqe_plan_enqueue_finalize(aqe_plan* p, void* stream) {
    int rc = plan_is_current(p);
    if (rc != AQE_OK) return rc;
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    HIPCHK(p->ctx, launch_finalize(p->d_state, finalize_params(p), p->d_result, pick(p, stream)));
    return AQE_OK;
}

int aqe_plan_totals_len(const aqe_plan* p, uint32_t* n_doubles) {
    if (!p || !n_doubles) return AQE_ERR_INVALID;
    *n_doubles = p->totals.ok ? p->totals.slots * kVec : 0;
    return AQE_OK;
}

int aqe_plan_enqueue_sweep_totals(aqe_plan* p, double* dev_totals, void* stream) {
    int rc = plan_is_current(p);
    if (rc != AQE_OK) return rc;
    if (!p->totals.ok) return fail(p->ctx, AQE_ERR_UNSUPPORTED, "this plan has no batched (totals) form; use the per-round calls");
    if (!dev_totals) return fail(p->ctx, AQE_ERR_INVALID, "dev_totals is null");
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    p->lev_used = 0;
    return launch_form(p, p->totals, true, dev_totals, pick(p, stream));
}

int aqe_plan_enqueue_replay(aqe_plan* p, const double* dev_totals, void* stream) {
    int rc = plan_is_current(p);
    if (rc != AQE_OK) return rc;
    if (!p->totals.ok) return fail(p->ctx, AQE_ERR_UNSUPPORTED, "this plan has no batched (totals) form");
    if (!dev_totals) return fail(p->ctx, AQE_ERR_INVALID, "dev_totals is null");
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    const uint32_t R = static_cast<uint32_t>(p->rounds.size());
    HIPCHK(p->ctx, launch_replay(dev_totals, R, p->host.has_topup ? 1u : 0u, fold_params(p, false), finalize_params(p),
                                 p->d_state, p->d_result, pick(p, stream)));
    return AQE_OK;
}

// Side streams of the batched form, owned by the context and shared by its batches.  Three, not one per plan: the
// runtime multiplexes streams onto a handful of hardware queues, and a stream that waits on an event blocks every
// other stream sharing its queue.  Three lanes plus the caller's stream each get a queue of their own, and two
// kernels in flight are already enough for one query's hand-off tail to overlap the next query's sweep.
int ensure_lanes(aqe_ctx* c) {
    while (c->lanes.size() < kBatchLanes) {
        hipStream_t s = nullptr;
        HIPCHK(c, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        c->lanes.push_back(s);
    }
    return AQE_OK;
}

struct aqe_batch {
    aqe_ctx* ctx = nullptr;
    std::vector<aqe_plan*> plans;   // plan i runs on lane i % kBatchLanes of the context
    std::vector<hipEvent_t> swept;  // lane l's sweeps of this batch are enqueued up to here
    hipEvent_t reduced = nullptr;   // the caller's stream up to (and including) the collective
};

void aqe_batch_destroy(aqe_batch* b) {
    if (!b) return;
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        for (hipStream_t s : b->ctx->lanes) (void)hipStreamSynchronize(s);
    }
    for (hipEvent_t e : b->swept) (void)hipEventDestroy(e);
    if (b->reduced) (void)hipEventDestroy(b->reduced);
    delete b;
}

int aqe_batch_create(aqe_plan* const* plans, uint32_t n, aqe_batch** out) {
    if (!plans || !out || n == 0) return AQE_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        if (!plans[i] || !plans[i]->ctx || plans[i]->ctx != plans[0]->ctx) return AQE_ERR_INVALID;
        if (!plans[i]->totals.ok) return fail(plans[i]->ctx, AQE_ERR_UNSUPPORTED, "a plan of the batch has no batched (totals) form");
        for (uint32_t j = 0; j < i; ++j)
            if (plans[j] == plans[i]) return fail(plans[i]->ctx, AQE_ERR_INVALID, "a plan may appear once in a batch (its hand-off scratch is its own)");
    }
    aqe_ctx* c = plans[0]->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    int rc = ensure_lanes(c);
    if (rc != AQE_OK) return rc;
    std::unique_ptr<aqe_batch, void (*)(aqe_batch*)> b(new aqe_batch, aqe_batch_destroy);
    b->ctx = c;
    b->plans.assign(plans, plans + n);
    for (size_t l = 0; l < kBatchLanes; ++l) {
        hipEvent_t e = nullptr;
        HIPCHK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        b->swept.push_back(e);
    }
    HIPCHK(c, hipEventCreateWithFlags(&b->reduced, hipEventDisableTiming));
    *out = b.release();
    return AQE_OK;
}

int aqe_batch_enqueue_sweeps(aqe_batch* b, double* dev_totals, uint64_t row_stride) {
    if (!b || !dev_totals) return AQE_ERR_INVALID;
    aqe_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    for (size_t i = 0; i < b->plans.size(); ++i) {
        aqe_plan* p = b->plans[i];
        int rc = plan_is_current(p);
        if (rc != AQE_OK) return rc;
        if (row_stride < static_cast<uint64_t>(p->totals.slots) * kVec) return fail(c, AQE_ERR_INVALID, "row_stride shorter than a plan's totals");
        p->lev_used = 0;
        // (the plan's previous replay precedes this sweep in its lane: nothing to wait for)
        rc = launch_form(p, p->totals, true, dev_totals + i * row_stride, c->lanes[i % kBatchLanes]);
        if (rc != AQE_OK) return rc;
    }
    for (size_t l = 0; l < kBatchLanes; ++l) HIPCHK(c, hipEventRecord(b->swept[l], c->lanes[l]));
    return AQE_OK;
}

int aqe_batch_join(aqe_batch* b, void* stream) {
    if (!b) return AQE_ERR_INVALID;
    aqe_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t main_s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    for (size_t l = 0; l < kBatchLanes; ++l) HIPCHK(c, hipStreamWaitEvent(main_s, b->swept[l], 0));
    return AQE_OK;
}

int aqe_batch_enqueue_replays(aqe_batch* b, const double* dev_totals, uint64_t row_stride, void* stream) {
    if (!b || !dev_totals) return AQE_ERR_INVALID;
    aqe_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t main_s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    HIPCHK(c, hipEventRecord(b->reduced, main_s));
    for (size_t l = 0; l < kBatchLanes; ++l) HIPCHK(c, hipStreamWaitEvent(c->lanes[l], b->reduced, 0));  // every lane waits for the collective
    for (size_t i = 0; i < b->plans.size(); ++i) {
        aqe_plan* p = b->plans[i];
        int rc = plan_is_current(p);
        if (rc != AQE_OK) return rc;
        HIPCHK(c, launch_replay(dev_totals + i * row_stride, static_cast<uint32_t>(p->rounds.size()), p->host.has_topup ? 1u : 0u,
                                fold_params(p, false), finalize_params(p), p->d_state, p->d_result, c->lanes[i % kBatchLanes]));
    }
    return AQE_OK;
}

int aqe_batch_fetch(aqe_batch* b, aqe_result* out_n) {
    if (!b || !out_n) return AQE_ERR_INVALID;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    for (size_t i = 0; i < b->plans.size(); ++i) {
        int rc = plan_is_current(b->plans[i]);
        if (rc == AQE_OK) rc = fetch(b->plans[i], out_n + i, b->ctx->lanes[i % kBatchLanes]);
        if (rc != AQE_OK) return rc;
    }
    return AQE_OK;
}

namespace {
int grouped_args(aqe_ctx* c, const aqe_query* q, int group_column) {
    if (!q) return fail(c, AQE_ERR_INVALID, "null query");
    if (group_column != AQE_GROUP_REGION && group_column != AQE_GROUP_PRODUCT) return fail(c, AQE_ERR_INVALID, "group_column must be AQE_GROUP_REGION or AQE_GROUP_PRODUCT");
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    if (!(q->sample_percent > 0.0)) return fail(c, AQE_ERR_INVALID, "sample_percent must be positive");
    return AQE_OK;
}
}  // namespace

int aqe_group_key_range(aqe_ctx* c, int group_column, int32_t* key_min, int32_t* key_max) {
    if (!c) return AQE_ERR_INVALID;
    if (!key_min || !key_max) return fail(c, AQE_ERR_INVALID, "null argument");
    if (group_column != AQE_GROUP_REGION && group_column != AQE_GROUP_PRODUCT) return fail(c, AQE_ERR_INVALID, "group_column must be AQE_GROUP_REGION or AQE_GROUP_PRODUCT");
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    HIPCHK(c, hipSetDevice(c->device));
    *key_min = std::numeric_limits<int32_t>::max();  // an empty shard: the neutral elements of MIN / MAX
    *key_max = std::numeric_limits<int32_t>::min();
    if (c->n_local == 0) return AQE_OK;
    int rc = ensure_keys(c, group_column);
    if (rc != AQE_OK) return rc;
    *key_min = c->key_min[group_column - 1];
    *key_max = c->key_max[group_column - 1];
    return AQE_OK;
}

int aqe_grouped_enqueue_bins(aqe_ctx* c, const aqe_query* q, int group_column, int32_t key_min, uint32_t nbins, double* dev_bins, void* stream) {
    if (!c) return AQE_ERR_INVALID;
    int rc = grouped_args(c, q, group_column);
    if (rc != AQE_OK) return rc;
    if (!dev_bins || nbins == 0 || nbins > static_cast<uint32_t>(kMaxGroupBins)) return fail(c, AQE_ERR_INVALID, "dev_bins null or nbins outside 1..1024");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    aqe_plan* p = nullptr;
    rc = cached_plan(c, q, &p);
    if (rc != AQE_OK) return rc;
    if (p->host.is_random || p->host.is_clt || p->host.on_sorted || p->rounds.size() > 1)
        return fail(c, AQE_ERR_UNSUPPORTED, "grouped reduction takes a single-round family sampler (exact, stride, rowid-mod, block, page, pointer, region ...)");
    for (const DevFamily& f : p->h_fams)
        if (f.flags & AQE_F_PAIR) return fail(c, AQE_ERR_UNSUPPORTED, "grouped reduction does not take pair families");
    if (p->rounds.empty() || c->n_local == 0 || p->rounds[0].ntiles == 0) {  // nothing of the sample in this shard
        HIPCHK(c, hipMemsetAsync(dev_bins, 0, static_cast<size_t>(nbins) * 4 * sizeof(double), s));
        return AQE_OK;
    }
    rc = ensure_keys(c, group_column);
    if (rc != AQE_OK) return rc;
    const int k = group_column - 1;
    if (c->key_min[k] < key_min || static_cast<int64_t>(c->key_max[k]) - key_min >= static_cast<int64_t>(nbins))
        return fail(c, AQE_ERR_INVALID, "this shard has keys outside [key_min, key_min + nbins)");
    const LaunchDesc& L = p->rounds[0];
    const unsigned grid = grouped_grid(L.ntiles);
    const size_t need = static_cast<size_t>(grid) * nbins * 4 * sizeof(double);
    if (c->grp_partial_bytes < need) {  // scratch lives with the context: no allocation on the query path after the first call
        if (c->grp_partial) (void)hipFree(c->grp_partial);
        c->grp_partial = nullptr;
        c->grp_partial_bytes = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->grp_partial), need));
        c->grp_partial_bytes = need;
    }
    HIPCHK(c, launch_grouped(sweep_common(p, p->d_fams + L.fam_offset, L.nfam), L.ntiles, c->keycol[k], key_min, nbins, c->grp_partial, grid, s));
    HIPCHK(c, launch_grouped_sum(c->grp_partial, grid, nbins, dev_bins, s));
    return AQE_OK;
}

int aqe_grouped_finish(aqe_ctx* c, const aqe_query* q, int32_t key_min, uint32_t nbins, const double* dev_bins, void* stream,
                       aqe_group_result* out, uint32_t cap, uint32_t* n_groups) {
    if (!c) return AQE_ERR_INVALID;
    if (!q || !n_groups || (cap && !out) || !dev_bins || nbins == 0 || nbins > static_cast<uint32_t>(kMaxGroupBins)) return fail(c, AQE_ERR_INVALID, "bad argument");
    if (q->agg < AQE_SUM || q->agg > AQE_COUNT || !(q->sample_percent > 0.0)) return fail(c, AQE_ERR_INVALID, "bad agg or sample_percent");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    *n_groups = 0;
    if (!c->grp_out) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->grp_out), kMaxGroupBins * sizeof(aqe_group_result)));
    std::vector<aqe_group_result> host(nbins);
    HIPCHK(c, launch_grouped_finish(dev_bins, nbins, key_min, c->shift, q->sample_percent, q->agg, c->grp_out, s));
    HIPCHK(c, hipMemcpyAsync(host.data(), c->grp_out, nbins * sizeof(aqe_group_result), hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    uint32_t g = 0;
    for (const aqe_group_result& r : host) {
        if (r.visited == 0) continue;  // a key nobody sampled
        if (g < cap) out[g] = r;
        ++g;
    }
    *n_groups = g;
    if (g > cap) return fail(c, AQE_ERR_CAPACITY, "more groups than the caller's buffer holds (n_groups has the count)");
    return AQE_OK;
}

int aqe_reduce_grouped(aqe_ctx* c, const aqe_query* q, int group_column, aqe_group_result* out, uint32_t cap, uint32_t* n_groups) {
    if (!c) return AQE_ERR_INVALID;
    if (!n_groups || (cap && !out)) return fail(c, AQE_ERR_INVALID, "null argument");
    int rc = grouped_args(c, q, group_column);
    if (rc != AQE_OK) return rc;
    *n_groups = 0;
    int32_t kmin = 0, kmax = -1;
    rc = aqe_group_key_range(c, group_column, &kmin, &kmax);
    if (rc != AQE_OK) return rc;
    if (kmax < kmin) return AQE_OK;  // empty table: no groups
    const int64_t span = static_cast<int64_t>(kmax) - kmin + 1;
    if (span > kMaxGroupBins) return fail(c, AQE_ERR_UNSUPPORTED, "group column spans more than 1024 distinct values");
    const uint32_t nbins = static_cast<uint32_t>(span);
    if (!c->grp_bins) HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->grp_bins), static_cast<size_t>(kMaxGroupBins) * 4 * sizeof(double)));
    rc = aqe_grouped_enqueue_bins(c, q, group_column, kmin, nbins, c->grp_bins, c->stream);
    if (rc != AQE_OK) return rc;
    return aqe_grouped_finish(c, q, kmin, nbins, c->grp_bins, c->stream, out, cap, n_groups);
}

int aqe_plan_enqueue_all(aqe_plan* p, void* stream) {
    int rc = plan_is_current(p);
    if (rc != AQE_OK) return rc;
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    return enqueue_all(p, pick(p, stream), p->profile);
}

int aqe_plan_fetch(aqe_plan* p, aqe_result* out, void* stream) {
    int rc = plan_is_current(p);
    if (rc != AQE_OK) return rc;
    if (!out) return AQE_ERR_INVALID;
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    return fetch(p, out, pick(p, stream));
}

int aqe_plan_last_kernel_ms(aqe_plan* p, float* ms) {
    if (!p || !ms) return AQE_ERR_INVALID;
    if (!p->timed) return fail(p->ctx, AQE_ERR_INVALID, "no timed execution yet");
    HIPCHK(p->ctx, hipEventSynchronize(p->ev1));
    HIPCHK(p->ctx, hipEventElapsedTime(ms, p->ev0, p->ev1));
    return AQE_OK;
}

int aqe_plan_set_profiling(aqe_plan* p, int enable) {
    if (!p) return AQE_ERR_INVALID;
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    p->profile = enable != 0;
    const size_t want = 2 * (p->rounds.size() + 2);
    while (p->profile && p->lev.size() < want) {
        hipEvent_t e;
        HIPCHK(p->ctx, hipEventCreate(&e));
        p->lev.push_back(e);
    }
    p->lev_used = 0;
    return AQE_OK;
}

int aqe_plan_launch_ms(aqe_plan* p, float* ms, uint32_t cap, uint32_t* n_out) {
    if (!p || !n_out) return AQE_ERR_INVALID;
    HIPCHK(p->ctx, hipSetDevice(p->ctx->device));
    *n_out = p->lev_used;
    if (!ms) return AQE_OK;
    if (cap < p->lev_used) return fail(p->ctx, AQE_ERR_CAPACITY, "launch_ms buffer too small");
    for (uint32_t i = 0; i < p->lev_used; ++i) {
        HIPCHK(p->ctx, hipEventSynchronize(p->lev[2 * i + 1]));
        HIPCHK(p->ctx, hipEventElapsedTime(&ms[i], p->lev[2 * i], p->lev[2 * i + 1]));
    }
    return AQE_OK;
}

int aqe_plan_launch_samples(const aqe_plan* p, uint64_t* samples, uint32_t cap, uint32_t* n_out) {
    if (!p || !n_out) return AQE_ERR_INVALID;
    // reports the launches of the form the plan last executed with (or will: the fused path by default)
    const int form = p->last_exec ? p->last_exec : (p->persist ? 1 : 0);
    if (form == 2) {  // batched multi-GPU form: one launch sweeps every slot, top-up included
        *n_out = 1;
        if (!samples) return AQE_OK;
        if (cap < 1) return AQE_ERR_CAPACITY;
        samples[0] = p->totals.samples;
        return AQE_OK;
    }
    const uint32_t sweeps = form == 1 ? 1u : static_cast<uint32_t>(p->rounds.size());
    const uint32_t n = sweeps + (p->host.has_topup ? 1u : 0u);
    *n_out = n;
    if (!samples) return AQE_OK;
    if (cap < n) return AQE_ERR_CAPACITY;
    if (form == 1) samples[0] = p->decide.samples;
    else for (size_t i = 0; i < p->rounds.size(); ++i) samples[i] = p->rounds[i].samples;
    if (p->host.has_topup) samples[sweeps] = p->topup.samples;
    return AQE_OK;
}

// ---- one-call forms -----------------------------------------------------------------------------
namespace {
// Synchronous callers (aqe_reduce, aqe_gather).  A plan of very many rounds — the reference's own cadence, ten rows
// per worker and round — would enqueue tens of thousands of launches of which all but the first few are device-side
// no-ops once should_stop is set.  Here the host enqueues a chunk of rounds, looks at should_stop, and stops
// launching when it is set.  (aqe_plan_enqueue_all stays fully asynchronous: it enqueues every round.)
constexpr uint32_t kSyncChunkRounds = 256;

int run_sync(aqe_plan* p, hipStream_t s, bool timed) {
    aqe_ctx* c = p->ctx;
    if (p->persist || p->rounds.size() <= kSyncChunkRounds) return enqueue_all(p, s, timed);
    if (timed) HIPCHK(c, hipEventRecord(p->ev0, s));
    p->lev_used = 0;
    p->last_exec = 0;
    const uint32_t R = static_cast<uint32_t>(p->rounds.size());
    for (uint32_t i = 0; i < R; i += kSyncChunkRounds) {
        for (uint32_t j = i; j < std::min(R, i + kSyncChunkRounds); ++j) {
            int rc = enqueue_launch(p, p->rounds[j], j, false, true, nullptr, s);
            if (rc != AQE_OK) return rc;
        }
        if (!p->host.is_clt) continue;
        // (the result block is pinned host memory the caller has not been handed yet: borrow a word of it)
        int32_t* peek = &p->h_result->device_status;
        HIPCHK(c, hipMemcpyAsync(peek, &p->d_state->stop, sizeof(int32_t), hipMemcpyDeviceToHost, s));
        HIPCHK(c, hipStreamSynchronize(s));
        if (*peek) break;  // every later round would leave at once: do not launch them
    }
    if (p->host.has_topup) {
        int rc = enqueue_launch(p, p->topup, R, true, true, nullptr, s);
        if (rc != AQE_OK) return rc;
    } else {
        // the last round of the plan carries the finalize; after an early break nobody has written the result
        HIPCHK(c, launch_finalize(p->d_state, finalize_params(p), p->d_result, s));
    }
    if (timed) HIPCHK(c, hipEventRecord(p->ev1, s));
    p->timed = timed;
    return AQE_OK;
}
}  // namespace

int aqe_reduce(aqe_ctx* c, const aqe_query* q, aqe_result* out) {
    if (!c || !q || !out) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->shard_lo != 0 || c->n_local != c->n_global)
        return fail(c, AQE_ERR_UNSUPPORTED, "aqe_reduce needs the whole table in this context; use the stepwise plan API for shards");
    aqe_plan* p = nullptr;
    int rc = cached_plan(c, q, &p);
    if (rc != AQE_OK) return rc;
    rc = run_sync(p, c->stream, true);
    if (rc != AQE_OK) return rc;
    return fetch(p, out, c->stream);
}

int aqe_gather(aqe_ctx* c, const aqe_query* q, void* out_aos32, uint64_t cap, uint64_t* n_out) {
    if (!c || !q || !n_out) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n_local && !c->aos) return fail(c, AQE_ERR_UNSUPPORTED, "record-returning samplers need AQE_STAGE_KEEP_AOS");
    if (c->shard_lo != 0 || c->n_local != c->n_global) return fail(c, AQE_ERR_UNSUPPORTED, "aqe_gather needs the whole table in this context");
    if (q->method == AQE_M_EXACT) return fail(c, AQE_ERR_UNSUPPORTED, "EXACT has no record-returning form");
    aqe_plan* p = nullptr;
    int rc = cached_plan(c, q, &p);
    if (rc != AQE_OK) return rc;
    // which launches contributed: all of them, except that the CLT sampler stops at its converged round
    // and appends `topup` rows (DB.cpp:1031-1040) — both known only after running the reduction.
    uint32_t rounds_used = static_cast<uint32_t>(p->rounds.size());
    uint64_t topup_rows = 0;
    if (p->host.is_clt) {
        aqe_result r;
        rc = run_sync(p, c->stream, false);
        if (rc == AQE_OK) rc = fetch(p, &r, c->stream);
        if (rc != AQE_OK) return rc;
        rounds_used = static_cast<uint32_t>(r.rounds);
        topup_rows = r.topup;
    }
    uint64_t total = p->host.is_random ? p->host.random_idx.size() : 0;
    if (!p->host.is_random)
        for (uint32_t r = 0; r < rounds_used; ++r) total += p->rounds[r].samples;
    total += topup_rows;
    *n_out = total;
    if (total == 0) return AQE_OK;
    if (!out_aos32 || cap < total) return fail(c, AQE_ERR_CAPACITY, "output buffer too small");
    aqe_record* d_out = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&d_out), total * sizeof(aqe_record)));
    hipError_t e = hipSuccess;
    if (p->host.is_random) {
        e = launch_gather_indexed(c->aos, c->shard_lo, p->d_idx, p->host.random_idx.size(), d_out, c->stream);
    } else {
        for (uint32_t r = 0; r < rounds_used && e == hipSuccess; ++r) {
            const LaunchDesc& L = p->rounds[r];
            if (L.nfam) e = launch_gather(c->aos, c->shard_lo, p->d_fams + L.fam_offset, L.nfam, L.ntiles, d_out, c->dense16,
                                          p->host.on_sorted ? c->sorted_row : nullptr, c->stream);
        }
        if (e == hipSuccess && topup_rows) {
            // the top-up is one strided family from row 0; keep its first `topup_rows` ordinals and place
            // them after the rows of the rounds that ran
            std::vector<DevFamily> tf(p->h_fams.begin() + static_cast<long>(p->topup.fam_offset),
                                      p->h_fams.begin() + static_cast<long>(p->topup.fam_offset + p->topup.nfam));
            uint64_t pos = total - topup_rows;
            for (auto& f : tf) {
                f.ord_hi = std::min<uint64_t>(f.ord_hi, std::max<uint64_t>(topup_rows, f.ord_lo));
                f.out_begin = pos;
                f.flags = 0;
            }
            DevFamily* d_tf = nullptr;
            e = hipMalloc(reinterpret_cast<void**>(&d_tf), tf.size() * sizeof(DevFamily));
            if (e == hipSuccess) e = hipMemcpy(d_tf, tf.data(), tf.size() * sizeof(DevFamily), hipMemcpyHostToDevice);
            if (e == hipSuccess) e = launch_gather(c->aos, c->shard_lo, d_tf, static_cast<uint32_t>(tf.size()), p->topup.ntiles, d_out, c->dense16, nullptr, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
            if (d_tf) (void)hipFree(d_tf);
        }
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_aos32, d_out, total * sizeof(aqe_record), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(c, AQE_ERR_HIP, std::string("gather: ") + hipGetErrorString(e));
    return AQE_OK;
}

}  // extern "C"
