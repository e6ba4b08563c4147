// capi.hip — the remaining entry points of include/aqe_hip.h: the one-call reducers (aqe_reduce, aqe_gather,
// aqe_reduce_grouped and its multi-GPU form) and the host-only helpers (planner access, WHERE parsing, status text).
// The table lives in table.hip, planned queries in plans.hip; the kernels in kernels.hip, persist.hip, grouped.hip.
#include "host.hpp"

using namespace aqe;

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

// ---- device scratch for hosts without a HIP runtime of their own -----------------------------------
int aqe_device_malloc(aqe_ctx* c, size_t bytes, void** out) {
    if (!c || !out) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMalloc(out, bytes ? bytes : 8));
    HIPCHK(c, hipMemset(*out, 0, bytes ? bytes : 8));
    HIPCHK(c, hipDeviceSynchronize());  // (the memset may still be in flight, on the null stream: the caller's streams do not wait for it)
    return AQE_OK;
}

int aqe_device_free(aqe_ctx* c, void* p) {
    if (!c) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    if (p) HIPCHK(c, hipFree(p));
    return AQE_OK;
}

int aqe_device_read(aqe_ctx* c, void* dst, const void* src, size_t bytes, void* stream) {
    if (!c || !dst || !src) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return AQE_OK;
}

int aqe_device_write(aqe_ctx* c, void* dst, const void* src, size_t bytes, void* stream) {
    if (!c || !dst || !src) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    HIPCHK(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    HIPCHK(c, hipStreamSynchronize(s));  // (the host buffer may be reused at once)
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
    if (P.is_random || P.is_perm) src = &none;
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

int aqe_plan_row_list(const aqe_query* q, uint64_t n_global, uint64_t shard_lo, uint64_t shard_hi, uint64_t* out, uint64_t cap, uint64_t* n_out) {
    if (!q) return AQE_ERR_INVALID;
    HostPlan P;
    std::string err;
    int rc = build_plan(*q, n_global, ClipWindow{shard_lo, shard_hi}, P, err);
    if (rc != AQE_OK) return fail(nullptr, rc, err);
    if (!P.is_random) return fail(nullptr, AQE_ERR_INVALID, "this sampler is a family sampler: use aqe_plan_families");
    if (n_out) *n_out = P.random_idx.size();
    if (out) {
        if (cap < P.random_idx.size()) return fail(nullptr, AQE_ERR_CAPACITY, "index buffer too small");
        std::copy(P.random_idx.begin(), P.random_idx.end(), out);
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

namespace {
// The sweep of a grouped reduction: this shard's sampled rows binned per workgroup into c->grp_partial
// ([*grid][nbins][4]); *grid == 0 when nothing of the sample lies in this shard.
int grouped_sweep(aqe_ctx* c, const aqe_query* q, int group_column, int32_t key_min, uint32_t nbins, hipStream_t s, unsigned* grid_out, const GroupFuse* fuse = nullptr) {
    *grid_out = 0;
    aqe_plan* p = nullptr;
    int rc = cached_plan(c, q, &p);
    if (rc != AQE_OK) return rc;
    if (p->host.is_random || p->host.is_perm || p->host.is_clt || p->host.on_sorted || p->rounds.size() > 1)
        return fail(c, AQE_ERR_UNSUPPORTED, "grouped reduction takes a single-round family sampler (exact, stride, rowid-mod, block, page, pointer, region ...)");
    for (const DevFamily& f : p->h_fams)
        if (f.flags & AQE_F_PAIR) return fail(c, AQE_ERR_UNSUPPORTED, "grouped reduction does not take pair families");
    if (p->rounds.empty() || c->n_local == 0 || p->rounds[0].ntiles == 0) return AQE_OK;  // nothing of the sample in this shard
    rc = ensure_keys(c, group_column);
    if (rc != AQE_OK) return rc;
    const int k = group_column - 1;
    if (c->key_min[k] < key_min || static_cast<int64_t>(c->key_max[k]) - key_min >= static_cast<int64_t>(nbins))
        return fail(c, AQE_ERR_INVALID, "this shard has keys outside [key_min, key_min + nbins)");
    // A strided sample is laid out over the stride-major view of the column (plans.hip): the keys are then read from the
    // key column's view in the same slot order — 12 bytes per sampled row instead of a whole line of each column.
    const int32_t* keys = c->keycol[k];
    if (p->view_rounds) {
        rc = ensure_key_view(c, group_column, p->view_step_rounds, &keys);
        if (rc != AQE_OK) return rc;
    }
    const LaunchDesc& L = p->rounds[0];
    const unsigned grid = grouped_grid(L.ntiles);
    const size_t need = fuse ? 0 : static_cast<size_t>(grid) * nbins * 4 * sizeof(double);  // (the fused form adds into its accumulator)
    if (c->grp_partial_bytes < need) {  // scratch lives with the context: no allocation on the query path after the first call
        if (c->grp_partial) (void)hipFree(c->grp_partial);
        c->grp_partial = nullptr;
        c->grp_partial_bytes = 0;
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->grp_partial), need));
        c->grp_partial_bytes = need;
    }
    HIPCHK(c, launch_grouped(sweep_common(p, p->d_fams + L.fam_offset, L.nfam), L.ntiles, keys, key_min, nbins, c->grp_partial, grid, s, fuse));
    *grid_out = grid;
    return AQE_OK;
}

// results leave through a pinned host buffer the device writes directly (no copy launch on the way out)
int ensure_group_out(aqe_ctx* c) {
    if (c->grp_out) return AQE_OK;
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->grp_out_host), kMaxGroupBins * sizeof(aqe_group_result), hipHostMallocMapped | hipHostMallocCoherent));
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->grp_out), c->grp_out_host, 0));
    return AQE_OK;
}

// scratch of the fused form: accumulator and tickets (device, zeroed once: every launch leaves them at zero), check words (pinned)
int ensure_group_fuse(aqe_ctx* c) {
    if (c->grp_acc) return AQE_OK;
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&c->grp_check_host), kMaxGroupBins * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(c->grp_check_host, 0, kMaxGroupBins * sizeof(unsigned long long));
    HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&c->grp_check), c->grp_check_host, 0));
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&c->grp_ticket), sizeof(unsigned) * kCounterWords));
    HIPCHK(c, hipMemset(c->grp_ticket, 0, sizeof(unsigned) * kCounterWords));
    double* acc = nullptr;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&acc), sizeof(double) * 4 * kMaxGroupBins));
    HIPCHK(c, hipMemset(acc, 0, sizeof(double) * 4 * kMaxGroupBins));
    HIPCHK(c, hipDeviceSynchronize());  // (the memsets run on the null stream, which the context's stream does not wait for)
    c->grp_acc = acc;
    return AQE_OK;
}

int collect_groups(aqe_ctx* c, uint32_t nbins, aqe_group_result* out, uint32_t cap, uint32_t* n_groups) {
    uint32_t g = 0;
    for (uint32_t b = 0; b < nbins; ++b) {
        const aqe_group_result& r = c->grp_out_host[b];
        if (r.visited == 0) continue;  // a key nobody sampled
        if (g < cap) out[g] = r;
        ++g;
    }
    *n_groups = g;
    if (g > cap) return fail(c, AQE_ERR_CAPACITY, "more groups than the caller's buffer holds (n_groups has the count)");
    return AQE_OK;
}
}  // namespace

int aqe_grouped_enqueue_bins(aqe_ctx* c, const aqe_query* q, int group_column, int32_t key_min, uint32_t nbins, double* dev_bins, void* stream) {
    if (!c) return AQE_ERR_INVALID;
    int rc = grouped_args(c, q, group_column);
    if (rc != AQE_OK) return rc;
    if (!dev_bins || nbins == 0 || nbins > static_cast<uint32_t>(kMaxGroupBins)) return fail(c, AQE_ERR_INVALID, "dev_bins null or nbins outside 1..1024");
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    unsigned grid = 0;
    rc = grouped_sweep(c, q, group_column, key_min, nbins, s, &grid);
    if (rc != AQE_OK) return rc;
    if (grid == 0) HIPCHK(c, hipMemsetAsync(dev_bins, 0, static_cast<size_t>(nbins) * 4 * sizeof(double), s));
    else HIPCHK(c, launch_grouped_sum(c->grp_partial, grid, nbins, dev_bins, s));
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
    int rc = ensure_group_out(c);
    if (rc != AQE_OK) return rc;
    HIPCHK(c, launch_grouped_finish(dev_bins, nbins, key_min, query_shift(c, *q), q->sample_percent, q->agg, c->grp_out, s));
    HIPCHK(c, hipStreamSynchronize(s));
    return collect_groups(c, nbins, out, cap, n_groups);
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
    if (q->agg < AQE_SUM || q->agg > AQE_COUNT) return fail(c, AQE_ERR_INVALID, "bad agg");
    HIPCHK(c, hipSetDevice(c->device));
    // A world of one needs no all-reduce between the sums and the estimates: ONE launch — the sweep, whose last workgroup
    // adds the bins up and works every group out, straight into the pinned result buffer — and the host polls the groups'
    // check words instead of waiting for the stream (AQE_GROUP_UNFUSED=1: the sweep, a second launch, a stream wait).
    rc = ensure_group_out(c);
    if (rc == AQE_OK) rc = ensure_group_fuse(c);
    if (rc != AQE_OK) return rc;
    static const bool unfused = std::getenv("AQE_GROUP_UNFUSED") != nullptr;
    unsigned grid = 0;
    if (unfused) {
        rc = grouped_sweep(c, q, group_column, kmin, nbins, c->stream, &grid);
        if (rc != AQE_OK) return rc;
        if (grid == 0) return AQE_OK;  // nothing sampled: no groups
        HIPCHK(c, launch_grouped_sum_finish(c->grp_partial, grid, nbins, kmin, query_shift(c, *q), q->sample_percent, q->agg, c->grp_out, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        return collect_groups(c, nbins, out, cap, n_groups);
    }
    GroupFuse f;
    f.acc = c->grp_acc; f.ticket = c->grp_ticket; f.out = c->grp_out; f.check = c->grp_check;
    f.epoch = c->epoch++;
    f.shift = query_shift(c, *q); f.pct = q->sample_percent; f.agg = q->agg;
    rc = grouped_sweep(c, q, group_column, kmin, nbins, c->stream, &grid, &f);
    if (rc != AQE_OK) return rc;
    if (grid == 0) return AQE_OK;  // nothing sampled: no groups
    {   // every bin's group has landed when its check word agrees with its fields (the stores may arrive in any order)
        const auto t0 = std::chrono::steady_clock::now();
        const volatile unsigned long long* chk = c->grp_check_host;
        uint32_t b = 0;
        for (unsigned spins = 0; b < nbins; ++spins) {
            aqe_group_result snap;
            const volatile unsigned long long* src = reinterpret_cast<const volatile unsigned long long*>(c->grp_out_host + b);
            unsigned long long w[sizeof(aqe_group_result) / 8];
            for (size_t i = 0; i < sizeof(aqe_group_result) / 8; ++i) w[i] = src[i];
            std::memcpy(&snap, w, sizeof snap);
            if (chk[b] == group_check(snap, f.epoch)) { ++b; continue; }
            if ((spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        if (b < nbins) HIPCHK(c, hipStreamSynchronize(c->stream));  // (never seen; the launch's end says the same)
    }
    return collect_groups(c, nbins, out, cap, n_groups);
}

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
    aqe_query in_place = *q;
    in_place.flags |= AQE_Q_NO_LAYOUT;  // records are gathered from the 32-byte rows: the families must address rows, not view slots
    int rc = cached_plan(c, &in_place, &p);
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
    uint64_t total = p->host.is_random ? p->host.random_idx.size() : p->host.is_perm ? p->host.perm_target : 0;
    if (!p->host.is_random && !p->host.is_perm)
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
    } else if (p->host.is_perm) {  // rows come back in draw order (k = 0, 1, ...): a sample, not a sorted set
        e = launch_gather_permuted(c->aos, perm_spec(p->host.perm_n, p->host.perm_lo, p->host.perm_target, p->host.perm_seed), d_out, c->stream);
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
