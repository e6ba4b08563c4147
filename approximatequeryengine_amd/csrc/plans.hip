// plans.hip — planned queries: families -> launch descriptors, the persistent-sweep forms, enqueue / fetch, the
// stepwise and batched multi-GPU entry points, batches of plans.  Host code.
#include "host.hpp"

using namespace aqe;

namespace aqe {

void destroy_plan(aqe_plan* p, bool device_idle) {
    if (!p) return;
    if (!device_idle) (void)hipDeviceSynchronize();  // fetch() may have returned before the plan's last launch had ended
    if (p->ctx && p->table_epoch == p->ctx->table_epoch) {  // (a replaced table took its views with it)
        if (p->view_step_rounds) release_stride_view(p->ctx, p->view_step_rounds, p->cached);
        if (p->view_step_topup) release_stride_view(p->ctx, p->view_step_topup, p->cached);
    }
    p->view_step_rounds = p->view_step_topup = 0;
    if (p->d_fams && p->d_fams != p->d_fams_small) (void)hipFree(p->d_fams);
    if (p->d_idx) (void)hipFree(p->d_idx);
    if (p->ctx && p->d_fams_small && p->partials && p->counter && p->d_state && p->h_result && p->ev0 && p->ev1 && p->d_ctl && p->d_rehearsal &&
        p->ctx->scratch_pool.size() < 128) {
        // (the device is idle — synchronised above — so the counters are back at zero and nothing is in flight)
        p->ctx->scratch_pool.push_back(PlanScratch{p->partials, p->counter, p->d_state, p->h_result, p->d_result, p->ev0, p->ev1, p->d_ctl, p->d_rehearsal, p->d_fams_small});
        p->partials = nullptr; p->counter = nullptr; p->d_state = nullptr; p->h_result = nullptr; p->ev0 = p->ev1 = nullptr;
        p->d_ctl = nullptr; p->d_rehearsal = nullptr; p->d_fams_small = nullptr;
    }
    if (p->partials) (void)hipFree(p->partials);
    if (p->counter) (void)hipFree(p->counter);
    if (p->d_ctl) (void)hipFree(p->d_ctl);
    if (p->d_rehearsal) (void)hipFree(p->d_rehearsal);
    if (p->d_fams_small) (void)hipFree(p->d_fams_small);
    for (SweepForm* f : {&p->decide, &p->totals, &p->head, &p->decide_lean, &p->totals_lean, &p->head_lean, &p->single_lean}) {
        if (f->d_ppart) (void)hipFree(f->d_ppart);  // (the form's family table lives in the same block)
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

namespace {

// Tile decomposition of one family window (kernels.hpp: one wave folds kTileOrdinals per tile).
void add_family(std::vector<DevFamily>& out, LaunchDesc& L, const aqe_family& f, uint64_t& out_pos, bool dense16, bool linear_ok = false) {
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
    if (linear_ok && s_lo != s_hi && tile == static_cast<uint64_t>(kTileOrdinals) && f.seg_len < tile && !(f.flags & AQE_F_PAIR)) {
        // many short segments (pages): tile j covers ordinals [(j_lo + j) tile, + tile) across segment boundaries
        d.flags |= kFamLinear;
        d.tiles_per_seg = 0;
        d.seg_lo = 0;
        d.j_lo = win_lo / tile;
        ntiles = (win_hi - 1) / tile + 1 - d.j_lo;
    } else if (s_lo == s_hi) {
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

}  // namespace

FoldParams fold_params(const aqe_plan* p, bool topup) {
    FoldParams f{};
    f.shift = query_shift(p->ctx, p->q);
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
    f.shift = query_shift(p->ctx, p->q);
    f.agg = p->q.agg;
    f.convention = p->q.convention;
    f.is_exact = p->q.method == AQE_M_EXACT;
    f.is_clt = p->host.is_clt;
    return f;
}

SweepCommon sweep_common(const aqe_plan* p, const DevFamily* fams, uint32_t nfam, bool topup) {
    const aqe_ctx* c = p->ctx;
    SweepCommon s{};
    const double* view = topup ? p->view_topup : p->view_rounds;
    s.amount = view ? view : (p->host.on_sorted ? c->sorted_amount : c->amount);
    s.shard_lo = view ? 0 : c->shard_lo;  // a view's families carry slot numbers, not rows
    s.fams = fams;
    s.nfam = nfam;
    s.has_where = p->q.has_where ? 1 : 0;
    s.wmin = p->q.where_min;
    s.wmax = p->q.where_max;
    s.shift = query_shift(c, p->q);
    s.dense16 = c->dense16 ? 1 : 0;
    // a query that sweeps more than the Infinity Cache holds finds nothing of it again: its loads go past the caches
    // (device_common.hpp, sweep_family<kNT>); smaller sweeps keep the default policy — the next execution, or a
    // neighbour in a batch, reads the same lines
    s.nt = p->nt ? 1 : 0;
    return s;
}

namespace {

// `index` is the launch's position in the query: rounds 0..R-1, then the top-up.  The first launch
// folds into a zeroed state (no memset), later CLT launches test should_stop on entry, and in the fused
// single-GPU form the last launch also writes the result.
RoundLaunch round_launch(const aqe_plan* p, const LaunchDesc& L, uint32_t index, bool topup, bool fused, double* out_vec, unsigned long long epoch = 0) {
    RoundLaunch a{};
    a.sw = sweep_common(p, p->d_fams ? p->d_fams + L.fam_offset : nullptr, L.nfam, topup);
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
    a.want_ticks = p->want_ticks ? 1u : 0u;
    if (epoch) {  // the host will poll for this execution's result (never for a captured launch: its arguments are frozen)
        a.epoch = epoch;
        a.result_seq = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p->d_result) + (reinterpret_cast<const volatile char*>(p->h_seq) - reinterpret_cast<const char*>(p->h_result)));
    }
    return a;
}

}  // namespace

int plan_is_current(aqe_plan* p) {
    if (!p || !p->ctx) return AQE_ERR_INVALID;
    if (p->table_epoch != p->ctx->table_epoch)
        return fail(p->ctx, AQE_ERR_INVALID, "plan was created for a table that has since been replaced");
    return AQE_OK;
}

namespace {

hipStream_t pick(aqe_plan* p, void* stream) { return stream ? static_cast<hipStream_t>(stream) : p->ctx->stream; }

int enqueue_launch(aqe_plan* p, const LaunchDesc& L, uint32_t index, bool topup, bool fused, double* out_vec, hipStream_t s, unsigned long long epoch = 0) {
    aqe_ctx* c = p->ctx;
    if (!topup && index == 0) { p->last_exec = 0; p->last_kernel = p->host.is_random ? AQE_KERNEL_INDEXED : p->host.is_perm ? AQE_KERNEL_PERMUTED : AQE_KERNEL_ROUND; }
    if (!epoch) p->poll_epoch = 0;  // this launch writes no check word: whatever it finishes is fetched the ordinary way
    RoundLaunch a = round_launch(p, L, index, topup, fused, out_vec, epoch);
    const bool prof = p->profile && 2 * (p->lev_used + 1) <= p->lev.size();
    hipEvent_t e0 = prof ? p->lev[2 * p->lev_used] : nullptr, e1 = prof ? p->lev[2 * p->lev_used + 1] : nullptr;
    if (p->host.is_random && !topup) HIPCHK(c, launch_indexed(a, p->d_idx, p->host.random_idx.size(), s, e0, e1));
    else if (p->host.is_perm && !topup)
        HIPCHK(c, launch_permuted(a, perm_spec(p->host.perm_n, p->host.perm_lo, p->host.perm_target, p->host.perm_seed), c->shard_lo, c->n_local, s, e0, e1));
    else HIPCHK(c, launch_round(a, s, e0, e1));
    if (prof) p->lev_used++;
    return AQE_OK;
}

// Lay the plan's rounds (optionally the top-up as one more slot) out as ONE tile list and work out which
// workgroups own tiles of which slot.
int build_sweep_form(aqe_plan* p, bool with_topup_slot, SweepForm& F, uint32_t grid, size_t nrounds) {
    aqe_ctx* c = p->ctx;
    std::vector<const LaunchDesc*> slots;
    for (size_t r = 0; r < nrounds; ++r) slots.push_back(&p->rounds[r]);
    F.grid = grid;
    F.more_rounds = nrounds < p->rounds.size() ? 1u : 0u;
    F.topup_slot = with_topup_slot && p->host.has_topup ? 1u : 0u;
    if (with_topup_slot && p->host.has_topup) slots.push_back(&p->topup);
    const size_t S = slots.size();
    uint64_t tiles = 0;
    // One launch has one column base (SweepCommon::amount), and the top-up may live in another copy of the column
    // than the rounds (its own stride-major view): its families' row numbers are then re-based — the distance
    // between the two allocations, in rows, is added (64-bit wrap-around; both are 256-byte aligned, so alignment
    // of every access is unchanged).
    const SweepCommon sw_r = sweep_common(p, nullptr, 0, false), sw_t = sweep_common(p, nullptr, 0, true);
    const int64_t base_gap = static_cast<int64_t>(reinterpret_cast<uintptr_t>(sw_t.amount) - reinterpret_cast<uintptr_t>(sw_r.amount)) / 8;  // rows
    const uint64_t topup_rebase = static_cast<uint64_t>(base_gap) - sw_t.shard_lo + sw_r.shard_lo;
    for (size_t r = 0; r < S; ++r) {
        const LaunchDesc& L = *slots[r];
        F.round_begin[r] = tiles;
        for (uint32_t i = 0; i < L.nfam; ++i) {
            DevFamily d = p->h_fams[L.fam_offset + i];
            if (slots[r] == &p->topup) { d.row0 += topup_rebase; d.row0_b += topup_rebase; }
            d.tile_begin += tiles;
            d.flags &= ~AQE_F_TOPUP;  // swept whole: the replay decides whether the top-up counts
            F.h_fams.push_back(d);
        }
        tiles += L.ntiles;
        F.samples += L.samples;
    }
    F.round_begin[S] = tiles;
    F.ntiles = tiles;
    {   // every family names its round and where that round ends
        size_t i = 0;
        for (size_t r = 0; r < S; ++r)
            for (uint32_t k = 0; k < slots[r]->nfam; ++k, ++i) { F.h_fams[i].round = static_cast<uint32_t>(r); F.h_fams[i].round_end = F.round_begin[r + 1]; }
    }
    F.slots = static_cast<uint32_t>(S);
    // Every wave but the monitor (wave 0 of workgroup 0) is a sweeper: sweeper v (physical wave v + 1) owns
    // tiles v, v + V, ...  The workgroups that own tiles of a slot form ONE cyclic run of workgroup ids (tiles
    // are consecutive, sweepers cyclic): find it by enumeration and insist on it — the monitor waits for
    // exactly these workgroups.
    const uint64_t G = grid, V = G * kPersistWaves - 1;
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
    // the monitor reads whole windows of kDecSteps steps: keep one window of slack behind the last slot.
    // Pad slots (a round's run rounded up to 8) are never written: zero data, flag word "always published".
    // ONE allocation and ONE copy per form: [partials, initialised][family table] (allocation calls are what a plan costs).
    const size_t ppart_words = static_cast<size_t>(kVec) * 8 * (static_cast<size_t>(F.step_begin[S]) + kDecSteps);
    const size_t fams_words = (F.h_fams.size() * sizeof(DevFamily) + 7) / 8;
    std::vector<uint64_t> init(ppart_words + fams_words, 0);
    for (size_t r = 0; r < S; ++r)
        for (size_t slot = 8 * static_cast<size_t>(F.step_begin[r]) + F.part_count[r]; slot < 8 * static_cast<size_t>(F.step_begin[r + 1]); ++slot)
            init[slot * kVec + 7] = kSlotAlways;
    if (!F.h_fams.empty()) std::memcpy(init.data() + ppart_words, F.h_fams.data(), F.h_fams.size() * sizeof(DevFamily));
    // The block reaches the device when the form is first launched (materialize_form): most plans run as lean launches and
    // never use their monitor-wave forms, and three allocations + copies were half of what creating a CLT plan cost.
    F.h_init = std::move(init);
    F.ppart_words = ppart_words;
    F.ok = true;
    return AQE_OK;
}

int materialize_form(aqe_ctx* c, SweepForm& F) {
    if (F.d_ppart || F.h_init.empty()) return AQE_OK;
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&F.d_ppart), F.h_init.size() * sizeof(uint64_t)));
    HIPCHK(c, hipMemcpy(F.d_ppart, F.h_init.data(), F.h_init.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    F.d_fams = F.h_fams.empty() ? nullptr : reinterpret_cast<DevFamily*>(reinterpret_cast<uint64_t*>(F.d_ppart) + F.ppart_words);  // (inside d_ppart's block: never freed on its own)
    std::vector<uint64_t>().swap(F.h_init);
    return AQE_OK;
}


// The lean variant of a form (lean.hip, k_sweep_lean): for a plan whose families are all plain runs of rows, or rows of
// equal blocks (block_sample: a SEGMENTED run).  Its own tile list — a run is tiled from its first row, a block from its
// first row — and its own slot list.  Several rounds: the tile list is cut into one contiguous share per workgroup and
// round r owns one slot per workgroup whose share holds tiles of it (consecutive workgroups; a round boundary splits at
// most one share, so there are fewer than workgroups + rounds slots whatever the size of the sweep).  ONE round (the
// single-round samplers: exact scans, strided samples through a view, blocks): tiles are dealt out wave by wave across
// the launch, every workgroup owns one slot, and the grid is no larger than one tile per wave needs (`exact_grid`: the
// caller — a batch — has fixed the group's size).  Leaves F.ok false (and returns AQE_OK) when the plan does not qualify.
int build_lean_form(aqe_plan* p, bool with_topup_slot, SweepForm& F, uint32_t grid, size_t nrounds, bool exact_grid = true) {
    aqe_ctx* c = p->ctx;
    static const bool off = [] { const char* e = std::getenv("AQE_LEAN"); return e && e[0] == '0'; }();  // diagnostics: AQE_LEAN=0
    if (off || !c->dense16 || grid == 0 || grid > static_cast<uint32_t>(kMaxPersistGrid)) return AQE_OK;
    std::vector<const LaunchDesc*> slots;
    for (size_t r = 0; r < nrounds; ++r) slots.push_back(&p->rounds[r]);
    const bool tslot = with_topup_slot && p->host.has_topup;
    if (tslot) slots.push_back(&p->topup);
    const size_t S = slots.size();
    if (S == 0 || S > static_cast<size_t>(kMaxPersistRounds)) return AQE_OK;
    const SweepCommon sw_r = sweep_common(p, nullptr, 0, false), sw_t = sweep_common(p, nullptr, 0, true);
    const int64_t base_gap = static_cast<int64_t>(reinterpret_cast<uintptr_t>(sw_t.amount) - reinterpret_cast<uintptr_t>(sw_r.amount)) / 8;  // rows (build_sweep_form)
    // up to kLeanMaxRuns runs travel in the launch descriptor (two per lane of a wave); up to kLeanWideRuns in a table of
    // their own in device memory (copied to LDS by every workgroup, searched by bisection)
    struct Run { uint64_t row0; uint32_t tile_begin, rows, meta; uint64_t pitch; uint32_t tps; };  // pitch != 0: segmented
    std::vector<Run> runs;
    uint64_t tiles = 0, samples = 0;
    bool any_seg = false;
    uint64_t round_begin[kMaxPersistRounds + 1] = {0};
    for (size_t r = 0; r < S; ++r) {
        const LaunchDesc& L = *slots[r];
        const bool is_topup = slots[r] == &p->topup;
        const SweepCommon& sw = is_topup ? sw_t : sw_r;
        round_begin[r] = tiles;
        for (uint32_t i = 0; i < L.nfam; ++i) {
            const DevFamily& d = p->h_fams[L.fam_offset + i];
            // one pointer, step 1, segments of at least half a tile
            if (!is_dense16(d.step, d.flags, d.seg_len) || (d.flags & kFamLinear) || d.ord_hi <= d.ord_lo || d.seg_len >= 0xffffffffull) return AQE_OK;
            const uint32_t meta = static_cast<uint32_t>(r) | (d.group != 0 ? kLeanMetaGroupB : 0u);
            // row of ordinal o: row0 + seg pitch - shard_lo + (o - seg seg_len)   (device_common.hpp, sweep_family; wraps like it)
            auto row_of = [&](uint64_t seg, uint64_t off) {
                uint64_t row = d.row0 + seg * d.pitch - sw.shard_lo + off;
                if (is_topup) row += static_cast<uint64_t>(base_gap);  // one launch has one column base: the rounds'
                return row;
            };
            auto plain = [&](uint64_t seg, uint64_t off, uint64_t len) {
                if (runs.size() == static_cast<size_t>(kLeanWideRuns) || len >= 0xffffffffull) return false;
                runs.push_back(Run{row_of(seg, off), static_cast<uint32_t>(tiles), static_cast<uint32_t>(len), meta, 0, 0});
                tiles += (len + kDenseTileOrdinals - 1) / kDenseTileOrdinals;
                samples += len;
                return true;
            };
            const uint64_t seg_lo = d.ord_lo / d.seg_len, seg_hi = (d.ord_hi - 1) / d.seg_len;
            if (seg_lo == seg_hi) {
                if (!plain(seg_lo, d.ord_lo - seg_lo * d.seg_len, d.ord_hi - d.ord_lo)) return AQE_OK;
                continue;
            }
            // a window over several blocks: what it cuts off the first and the last block are plain runs, the whole blocks
            // between them one segmented run
            uint64_t first_full = seg_lo, end_full = seg_hi + 1;  // [first_full, end_full)
            const uint64_t off_lo = d.ord_lo - seg_lo * d.seg_len, rows_hi = d.ord_hi - seg_hi * d.seg_len;
            if (off_lo) { if (!plain(seg_lo, off_lo, d.seg_len - off_lo)) return AQE_OK; ++first_full; }
            if (rows_hi < d.seg_len) --end_full;
            if (end_full > first_full) {
                const uint64_t nseg = end_full - first_full, tps = (d.seg_len + kDenseTileOrdinals - 1) / kDenseTileOrdinals;
                if (nseg == 1) {
                    if (!plain(first_full, 0, d.seg_len)) return AQE_OK;
                } else {
                    if (runs.size() == static_cast<size_t>(kLeanWideRuns) || nseg * tps >= 0xffffffffull) return AQE_OK;
                    runs.push_back(Run{row_of(first_full, 0), static_cast<uint32_t>(tiles), static_cast<uint32_t>(d.seg_len), meta | kLeanMetaSeg, d.pitch, static_cast<uint32_t>(tps)});
                    tiles += nseg * tps;
                    samples += nseg * d.seg_len;
                    any_seg = true;
                }
            }
            if (rows_hi < d.seg_len && !plain(seg_hi, 0, rows_hi)) return AQE_OK;
        }
        if (tiles == round_begin[r]) return AQE_OK;  // a slot without tiles: the other forms deal with it
    }
    round_begin[S] = tiles;
    const size_t nruns = runs.size();
    if (any_seg && nruns > static_cast<size_t>(kLeanMaxRuns) / 2) return AQE_OK;  // (a segmented run's geometry takes entry i + 64)
    if (tiles == 0 || tiles + grid >= 0xffffffffull) return AQE_OK;
    // one round: dealt out wave by wave, on no more workgroups than one tile per wave needs — every workgroup then has a tile
    uint64_t G = grid;
    bool dealt = S == 1;
    if (dealt) {
        const uint64_t need = (tiles + kPersistWaves - 1) / kPersistWaves;
        if (!exact_grid) G = std::min<uint64_t>(G, need);
        else if (need < G) dealt = false;
    }
    uint32_t part_first[kMaxPersistRounds] = {0};
    uint64_t K = 0;
    if (dealt) {
        F.slot_begin[1] = static_cast<uint32_t>(G);
    } else {
        K = (tiles + G - 1) / G;  // workgroup b owns the tiles [b K, (b + 1) K)
        for (size_t r = 0; r < S; ++r) {
            const uint64_t first = round_begin[r] / K, last = (round_begin[r + 1] - 1) / K;
            part_first[r] = static_cast<uint32_t>(first);
            F.slot_begin[r + 1] = F.slot_begin[r] + static_cast<uint32_t>(last - first + 1);
        }
    }
    if (F.slot_begin[S] > static_cast<uint32_t>(kLeanMaxSlots)) return AQE_OK;
    F.wide = nruns > static_cast<size_t>(kLeanMaxRuns);
    const size_t ppart_bytes = sizeof(double) * kVec * F.slot_begin[S];
    // the partial list: every slot is written by its workgroup in every launch, nothing to initialise.  (The run table
    // travels in the launch descriptor; a wide plan's sits behind the partials, in the same block.)
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&F.d_ppart), ppart_bytes + (F.wide ? sizeof(LeanWideRuns) : 0)));
    F.h_runs = LeanRuns{};
    for (int i = 0; i < kLeanMaxRuns; ++i) F.h_runs.tile_begin[i] = 0xffffffffu;
    F.nruns = static_cast<uint32_t>(nruns);
    if (F.wide) {
        auto w = std::make_unique<LeanWideRuns>();
        for (size_t i = 0; i < nruns; ++i) {
            const uint32_t r = runs[i].meta & 0xffu;
            w->row0[i] = runs[i].row0; w->tile_begin[i] = runs[i].tile_begin; w->rows[i] = runs[i].rows; w->meta[i] = runs[i].meta;
            w->slot[i] = F.slot_begin[r] | (part_first[r] << 16);
        }
        F.d_wide = reinterpret_cast<LeanWideRuns*>(reinterpret_cast<char*>(F.d_ppart) + ppart_bytes);
        HIPCHK(c, hipMemcpy(F.d_wide, w.get(), sizeof(LeanWideRuns), hipMemcpyHostToDevice));
    } else {
        for (size_t i = 0; i < nruns; ++i) {
            const uint32_t r = runs[i].meta & 0xffu;
            F.h_runs.row0[i] = runs[i].row0; F.h_runs.tile_begin[i] = runs[i].tile_begin; F.h_runs.rows[i] = runs[i].rows; F.h_runs.meta[i] = runs[i].meta;
            F.h_runs.slot[i] = F.slot_begin[r] | (part_first[r] << 16);
            if (runs[i].meta & kLeanMetaSeg) { F.h_runs.row0[i + 64] = runs[i].pitch; F.h_runs.meta[i + 64] = runs[i].tps; }
        }
    }
    F.lean = true;
    F.tiles_per_wg = static_cast<uint32_t>(K);  // 0: dealt
    F.slots = static_cast<uint32_t>(S);
    F.ntiles = tiles;
    F.samples = samples;
    F.grid = static_cast<uint32_t>(G);
    F.more_rounds = nrounds < p->rounds.size() ? 1u : 0u;
    F.topup_slot = tslot ? 1u : 0u;
    F.ok = true;
    return AQE_OK;
}

}  // namespace

namespace {

// Sweeps below 65 536 samples are launch-bound and gain no bandwidth from a copy of the column: they stay in place — unless
// the plan has several rounds, the copy is cheap (a shard of at most 16 M rows: 128 MB, built in well under a millisecond)
// and the sweep is at least a few tiles: a multi-round plan over views is a plan of plain runs, and those take the lean
// launch (10 us instead of 17 for a CLT query).
constexpr uint64_t kViewMinSamples = 1u << 16, kViewMinSamplesSmallTable = 1u << 12, kViewSmallTableRows = 1u << 24;
constexpr uint64_t kViewMaxStep = 1024;

// Rewrites a list of single-segment strided families of one common step into dense families over that step's
// stride-major view (a PAIR family becomes two: its pointers have different residues, i.e. two streams).
int families_to_view(aqe_ctx* c, std::vector<std::vector<aqe_family>*> lists, const double** view, uint64_t* view_step, bool multi_round) {
    uint64_t step = 0, total = 0;
    for (auto* L : lists)
        for (const aqe_family& f : *L) {
            if (f.pitch != 0 || f.step < 2 || f.step > kViewMaxStep || (step && f.step != step)) return AQE_OK;  // not this shape
            step = f.step;
            total += family_size(f);
        }
    if (!step || total < (multi_round && c->n_local <= kViewSmallTableRows ? kViewMinSamplesSmallTable : kViewMinSamples)) return AQE_OK;
    uint64_t M = 0, q0 = 0;
    int rc = ensure_stride_view(c, step, view, &M, &q0);
    if (rc != AQE_OK) return rc;
    if (!*view) return AQE_OK;  // the table holds its quota of views, all in use: this step stays in place
    *view_step = step;
    c->stride_views[step].refs++;  // released by destroy_plan
    auto slot0 = [&](uint64_t row0) { return (row0 % step) * M + row0 / step - q0; };  // wraps below the shard: the ordinal window brings it back
    for (auto* L : lists) {
        std::vector<aqe_family> out;
        for (const aqe_family& f : *L) {
            aqe_family a = f;
            a.flags &= ~AQE_F_PAIR;
            a.step = 1;
            a.row0 = slot0(f.row0);
            a.seg_len = std::max<uint64_t>(f.seg_len, f.ord_hi);
            a.row0_b = a.ord_lo_b = a.ord_hi_b = 0;
            if (a.ord_hi > a.ord_lo) out.push_back(a);
            if ((f.flags & AQE_F_PAIR) && f.ord_hi_b > f.ord_lo_b) {
                aqe_family b = a;
                b.group = 1;
                b.row0 = slot0(f.row0_b);
                b.ord_lo = f.ord_lo_b;
                b.ord_hi = f.ord_hi_b;
                b.seg_len = std::max<uint64_t>(f.seg_len, f.ord_hi_b);
                out.push_back(b);
            }
        }
        L->swap(out);
    }
    return AQE_OK;
}

}  // namespace

int create_plan(aqe_ctx* c, const aqe_query* q, aqe_plan** out, const GivenFamilies* given) {
    if (!c->staged) return fail(c, AQE_ERR_NO_TABLE, "no table staged");
    if (q->agg < AQE_SUM || q->agg > AQE_COUNT) return fail(c, AQE_ERR_INVALID, "agg must be AQE_SUM, AQE_AVG or AQE_COUNT");
    if (q->convention < AQE_EST_CLI || q->convention > AQE_EST_RAW) return fail(c, AQE_ERR_INVALID, "convention must be AQE_EST_CLI, AQE_EST_CPP or AQE_EST_RAW");
    if (q->has_where && (q->where_min != q->where_min || q->where_max != q->where_max)) return fail(c, AQE_ERR_INVALID, "WHERE bound is NaN");
    std::unique_ptr<aqe_plan, void (*)(aqe_plan*)> p(new aqe_plan(), [](aqe_plan* x) { destroy_plan(x); });
    p->ctx = c;
    p->q = *q;
    p->table_epoch = c->table_epoch;
    std::string err;
    const double* zone_var = nullptr;
    const bool whole_table = c->shard_lo == 0 && c->n_local == c->n_global;
    if (!given && q->method == AQE_M_ADAPTIVE_BLOCK) {
        // (a shard cannot see the other shards' zones: its ranks agree on the variances first, aqe_set_zone_variances)
        if (!whole_table && !c->zone_var_valid)
            return fail(c, AQE_ERR_UNSUPPORTED, "adaptive_block_sample over a sharded table: all-reduce aqe_zone_moments and hand the variances to aqe_set_zone_variances first");
        int rc0 = ensure_zone_variances(c);
        if (rc0 != AQE_OK) return rc0;
        zone_var = c->zone_var;
    }
    if (!given && q->method == AQE_M_STRATIFIED_BLOCK) {
        if (!whole_table)
            return fail(c, AQE_ERR_UNSUPPORTED, "stratified_block_sample over a sharded table: map the global sorted positions with aqe_sorted_counts and plan the local runs with aqe_plan_create_families");
        int rc0 = ensure_sorted(c);
        if (rc0 != AQE_OK) return rc0;
    }
    int rc = AQE_OK;
    if (given) {
        // the caller's families, as a single-round plan (aqe_plan_create_families)
        HostPlan& P = p->host;
        P = HostPlan{};
        P.pct = q->sample_percent;
        P.visible_rows = c->n_global;
        P.rounds = 1;
        P.round_fams.assign(1, {});
        P.global_samples = given->global_samples;
        P.on_sorted = given->on_sorted;
        if (given->on_sorted) {
            rc = ensure_sorted(c);
            if (rc != AQE_OK) return rc;
        }
        for (uint32_t i = 0; i < given->n; ++i) {
            aqe_family f = given->fams[i];
            if ((f.flags & (AQE_F_PAIR | AQE_F_TOPUP)) || f.group != 0) return fail(c, AQE_ERR_INVALID, "aqe_plan_create_families: plain families of group 0 only");
            if (f.ord_hi <= f.ord_lo) continue;
            if (f.seg_len == 0 || f.step == 0) return fail(c, AQE_ERR_INVALID, "aqe_plan_create_families: seg_len and step must be positive");
            const uint64_t last = f.ord_hi - 1, limit = given->on_sorted ? c->n_local : c->n_global;
            // (the last ordinal's row, with the products checked: a family that leaves the table must never reach a kernel)
            const uint64_t seg = last / f.seg_len, within = last % f.seg_len;
            if ((seg && f.pitch > (limit - 1) / seg) || (within && f.step > (limit - 1) / within)) return fail(c, AQE_ERR_INVALID, "aqe_plan_create_families: family leaves the table");
            const uint64_t off = seg * f.pitch + within * f.step;
            if (f.row0 >= limit || off > limit - 1 - f.row0) return fail(c, AQE_ERR_INVALID, "aqe_plan_create_families: family leaves the table");
            // (rows must ascend with the ordinal — segments in order, none overlapping — as every planned family's do: then the
            // last ordinal's row, checked above, is the largest, and clipping to a shard may bisect)
            if (seg && ((f.seg_len - 1) > (~0ull) / f.step || f.pitch <= (f.seg_len - 1) * f.step))
                return fail(c, AQE_ERR_INVALID, "aqe_plan_create_families: segments must not overlap (pitch > (seg_len - 1) * step)");
            if (given->on_sorted) {
                f.row0 += c->shard_lo;  // (device families carry global rows; the sorted column's positions are this shard's own)
                P.round_fams[0].push_back(f);
            } else {
                clip_family(P.round_fams[0], f, ClipWindow{c->shard_lo, c->shard_lo + c->n_local});
            }
        }
    } else {
        rc = build_plan(*q, c->n_global, ClipWindow{c->shard_lo, c->shard_lo + c->n_local}, p->host, err, zone_var);
    }
    if (rc != AQE_OK) return fail(c, rc, err);
    if (!(q->flags & AQE_Q_NO_LAYOUT) && c->n_local && !p->host.is_random && !p->host.is_perm && !p->host.on_sorted) {
        // strided pointers — a CLT query's rounds (one step) and its top-up (another), the strided samplers — read
        // stride-major views of the column: dense streams
        std::vector<std::vector<aqe_family>*> rounds;
        for (auto& rf : p->host.round_fams) rounds.push_back(&rf);
        const bool multi_round = p->host.round_fams.size() >= 2;
        rc = families_to_view(c, rounds, &p->view_rounds, &p->view_step_rounds, multi_round);
        if (rc == AQE_OK && p->host.has_topup) rc = families_to_view(c, {&p->host.topup_fams}, &p->view_topup, &p->view_step_topup, multi_round);
        if (rc != AQE_OK) return rc;
    }
    uint64_t out_pos = 0;
    for (const auto& rf : p->host.round_fams) {
        LaunchDesc L;
        L.fam_offset = p->h_fams.size();
        for (const auto& f : rf) add_family(p->h_fams, L, f, out_pos, c->dense16, !(q->flags & AQE_Q_NO_LAYOUT));
        p->rounds.push_back(L);
    }
    if (p->host.is_random) {
        LaunchDesc L;
        L.samples = p->host.random_idx.size();
        p->rounds.assign(1, L);
    }
    if (p->host.is_perm) {  // (an empty sample keeps no launch: the query is a zero state, finalized)
        LaunchDesc L;
        L.samples = p->host.perm_target;  // over the whole table; a shard folds the rows it holds
        if (p->host.perm_target) p->rounds.assign(1, L); else p->rounds.clear();
    }
    if (p->host.has_topup) {
        p->topup.fam_offset = p->h_fams.size();
        for (const auto& f : p->host.topup_fams) add_family(p->h_fams, p->topup, f, out_pos, c->dense16);
    }
    constexpr size_t kSeqOffset = (sizeof(aqe_result) + 63) / 64 * 64;  // the sequence word on its own cache line
    if (!c->scratch_pool.empty()) {  // a destroyed plan's scratch, as it is (host.hpp, PlanScratch)
        const PlanScratch sc = c->scratch_pool.back();
        c->scratch_pool.pop_back();
        p->partials = sc.partials; p->counter = sc.counter; p->d_state = sc.d_state;
        p->h_result = sc.h_result; p->d_result = sc.d_result; p->ev0 = sc.ev0; p->ev1 = sc.ev1;
        p->d_ctl = sc.d_ctl; p->d_rehearsal = sc.d_rehearsal; p->d_fams_small = sc.d_fams_small;
    } else {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->partials), sizeof(double) * kVec * kMaxBlocks));
        HIPCHK(c, hipMemsetAsync(p->partials, 0, sizeof(double) * kVec * kMaxBlocks, c->stream));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->counter), sizeof(unsigned) * kCounterWords));
        HIPCHK(c, hipMemsetAsync(p->counter, 0, sizeof(unsigned) * kCounterWords, c->stream));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_state), 1024));  // (QueryState; the rest is scratch of the ablation builds, tools/ab_ablate.sh)
        static_assert(sizeof(QueryState) <= 256, "state block");
        HIPCHK(c, hipMemsetAsync(p->d_state, 0, sizeof(QueryState), c->stream));
        // (coherent, i.e. fine-grained: the device's stores must reach host memory while the launch is still running — fetch() polls)
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void**>(&p->h_result), kSeqOffset + 64, hipHostMallocMapped | hipHostMallocCoherent));
        HIPCHK(c, hipHostGetDevicePointer(reinterpret_cast<void**>(&p->d_result), p->h_result, 0));
        HIPCHK(c, hipEventCreate(&p->ev0));
        HIPCHK(c, hipEventCreate(&p->ev1));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_ctl), sizeof(PersistCtl)));
        HIPCHK(c, hipMemsetAsync(p->d_ctl, 0, sizeof(PersistCtl), c->stream));
        HIPCHK(c, hipMalloc(&p->d_rehearsal, sizeof(QueryState) + sizeof(aqe_result)));
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_fams_small), kPoolFams * sizeof(DevFamily)));
        // (the tickets must be zero before the first launch — on whatever stream the caller picks — draws one: the memsets
        // run on the context's stream and only that stream is waited for; other work in flight on the device goes on)
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    std::memset(p->h_result, 0, kSeqOffset + 64);
    p->h_seq = reinterpret_cast<volatile unsigned long long*>(reinterpret_cast<char*>(p->h_result) + kSeqOffset);
    if (!p->h_fams.empty()) {  // (a short table goes into the pooled scratch: no allocation)
        if (p->h_fams.size() <= kPoolFams) p->d_fams = p->d_fams_small;
        else HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_fams), p->h_fams.size() * sizeof(DevFamily)));
        HIPCHK(c, hipMemcpy(p->d_fams, p->h_fams.data(), p->h_fams.size() * sizeof(DevFamily), hipMemcpyHostToDevice));
    }
    if (!p->host.random_idx.empty()) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&p->d_idx), p->host.random_idx.size() * sizeof(uint64_t)));
        HIPCHK(c, hipMemcpy(p->d_idx, p->host.random_idx.data(), p->host.random_idx.size() * sizeof(uint64_t),
                            hipMemcpyHostToDevice));
    }
    {   // (what one execution sweeps, in bytes of this shard: decides the load policy, sweep_common)
        uint64_t swept = p->host.has_topup ? p->topup.samples : 0;
        for (const LaunchDesc& L : p->rounds) swept += L.samples;
        p->nt = swept * sizeof(double) > kInfinityCacheBytes;
        if (const char* e = std::getenv("AQE_NT")) p->nt = e[0] == '1';  // diagnostics (tools/ab_nt.py): force the load policy, read per plan
    }
    {   // persistent single-launch forms of the rounds
        const size_t R = p->rounds.size();
        // A query that will run beside others (AQE_Q_SHARE_GPU) takes half the compute units: two such launches then sit
        // side by side on the chip and one's hand-off tail hides behind the other's sweep — 133 k against 107 k
        // aggregates/s with 32 in flight — at the price of a longer launch when it runs alone (19.4 against 16.4 us).
        p->grid = (q->flags & AQE_Q_SHARE_GPU) ? std::max(16u, c->persist_grid / 2) : c->persist_grid;
        const bool multi = !p->host.is_random && !p->host.is_perm && R >= 2 && p->grid > 0;
        {   // Will the query need (nearly) all of its rounds — the rule predicted to hold in the last round, within the one
            // round of margin of it, or not at all (the prediction described below)?  Then nothing is gained by judging early.
            p->predicted_full = !p->host.is_clt || q->max_error_percent <= 0.0;
            if (p->host.is_clt && c->head_cv > 0.0 && q->max_error_percent > 0.0) {
                const double root = p->host.clt.z * c->head_cv * 100.0 / q->max_error_percent;
                const double n_stop = std::max(50.0, root * root), workers = static_cast<double>(std::max(1, p->host.clt.n_workers));
                double leader = 0.0;
                size_t r_pred = 0;
                for (; r_pred < R && leader < n_stop; ++r_pred) leader += static_cast<double>(p->rounds[r_pred].samples) / workers;
                p->predicted_full = r_pred + 1 >= R;
            }
        }
        const bool whole = c->shard_lo == 0 && c->n_local == c->n_global;
        bool every_round_has_tiles = true;
        for (size_t r = 0; r < R; ++r) every_round_has_tiles = every_round_has_tiles && p->rounds[r].ntiles > 0;
        if (multi && whole && every_round_has_tiles && R <= static_cast<size_t>(kMaxPersistRounds) && !(q->flags & AQE_Q_NO_PERSIST)) {
            int rc2 = build_sweep_form(p.get(), false, p->decide, p->grid, R);
            if (rc2 == AQE_OK) rc2 = build_lean_form(p.get(), false, p->decide_lean, p->grid, R);
            if (rc2 != AQE_OK) return rc2;
            p->persist = true;
            // The single launch sweeps every round speculatively: right when the query runs (almost) to the end, a waste
            // when it stops early and most of the sweep would have been for nothing.  Which one applies depends on the
            // data, so it is PREDICTED — from the coefficient of variation of the table's head and the error rule
            // (DB.cpp:936-961: stop once z cv / sqrt(n) <= e/100) — a fixed function of table and query.  A query
            // predicted to stop early gets the HEAD form below.  AQE_Q_NO_PERSIST / AQE_Q_FORCE_PERSIST override.
            if (p->host.is_clt && !(q->flags & AQE_Q_FORCE_PERSIST) && c->head_cv > 0.0 && q->max_error_percent > 0.0) {
                const double root = p->host.clt.z * c->head_cv * 100.0 / q->max_error_percent;
                const double n_stop = std::max(50.0, root * root);  // samples the LEADER needs before rule A can hold (n >= 50, DB.cpp:958)
                const double workers = static_cast<double>(std::max(1, p->host.clt.n_workers));
                double swept = 0.0, leader = 0.0;
                size_t r_stop = 0;  // rounds swept when the rule is predicted to hold (a round gives every worker the same number of rows)
                for (; r_stop < R && leader < n_stop; ++r_stop) {
                    swept += static_cast<double>(p->rounds[r_stop].samples);
                    leader += static_cast<double>(p->rounds[r_stop].samples) / workers;
                }
                p->per_round = p->host.clt.n_fast > 0 && swept * 4.0 <= static_cast<double>(p->decide.samples);
                if (p->per_round) {
                    // The HEAD form: ONE launch that sweeps the predicted rounds plus one of margin (four times the rows:
                    // twice the predicted cv) on just enough workgroups for one tile per wave.  If the query has not
                    // stopped by then, the result says so (topup_pending == 2), fetch() launches the remaining rounds one
                    // by one, and the plan takes the full single launch from then on.
                    // A query that stops this early is short of rows and takes the reference's top-up (DB.cpp:1031-1040):
                    // the head form sweeps it along with the rounds, as one more slot, and the monitor adds it when due.
                    // (Measured, bench query at e = 1 %: 230 k aggregates/s and 14 us per launch, against 76-100 k for
                    // one launch per round replayed as a graph, which is what such plans used before.)
                    const size_t r_head = std::min(R, r_stop + 1);
                    p->r_head = r_head;
                    // Attempts, in order: rounds + top-up on one tile per wave; the same on half the largest grid when that many
                    // workgroups' partials would not fit the monitor's one window of steps and the top-up is small enough
                    // (<= 64 MB) for half the chip to sweep it faster than a second launch would start; the rounds alone.
                    struct Attempt { bool topup; uint32_t cap; };
                    std::vector<Attempt> attempts;
                    if (p->host.has_topup) {
                        attempts.push_back({true, p->grid});
                        if (p->grid > static_cast<uint32_t>(kMaxPersistGrid) / 2 && p->topup.samples * 8 <= (64ull << 20))
                            attempts.push_back({true, static_cast<uint32_t>(kMaxPersistGrid) / 2});
                    }
                    attempts.push_back({false, p->grid});
                    for (const Attempt& at : attempts) {
                        uint64_t tiles = at.topup ? p->topup.ntiles : 0;
                        for (size_t r = 0; r < r_head; ++r) tiles += p->rounds[r].ntiles;
                        uint32_t g = 1;
                        // one tile per sweeper wave: the sweep is a few microseconds of latency, not of bandwidth (measured, bench
                        // query at e = 1 %: 14 us per launch and 230 k aggregates/s so, 23 us and 173 k with two tiles per wave)
                        while (g < at.cap && static_cast<uint64_t>(g) * kPersistWaves < tiles) g *= 2;
                        SweepForm F;
                        rc2 = build_sweep_form(p.get(), at.topup, F, g, r_head);
                        // (with the top-up the monitor judges once, from one window of steps)
                        if (rc2 == AQE_OK && !(at.topup && F.step_begin[F.slots] > static_cast<uint32_t>(kDecSteps))) {
                            p->head = std::move(F);
                            // the lean variant has no window of steps to fit: it takes the top-up along whenever the plan has
                            // one, on the whole grid when the top-up is what it mostly sweeps
                            if (p->host.has_topup && !at.topup) rc2 = build_lean_form(p.get(), true, p->head_lean, p->grid, r_head);
                            if (rc2 == AQE_OK && !p->head_lean.ok) rc2 = build_lean_form(p.get(), at.topup, p->head_lean, g, r_head);
                            if (rc2 != AQE_OK) return rc2;
                            break;
                        }
                        if (F.d_ppart) (void)hipFree(F.d_ppart);
                        if (rc2 != AQE_OK) return rc2;
                    }
                }
            }
        }
        if (multi && R <= static_cast<size_t>(kMaxPersistRounds)) {
            int rc2 = build_sweep_form(p.get(), false, p->totals, p->grid, R);
            if (rc2 == AQE_OK) rc2 = build_lean_form(p.get(), false, p->totals_lean, p->grid, R);
            if (rc2 != AQE_OK) return rc2;
        }
        // ONE round — exact scans (DB.cpp:242-251), strided samples read through a view (DB.cpp:1526-1603), blocks
        // (DB.cpp:1151-1181) — as a lean launch of its own: the same sweep loop and hand-off as a CLT query's, tiles dealt out
        // wave by wave.  k_round keeps the rest (pages, strided samples in place) and the stepwise multi-GPU form.
        // (Below ~1 500 tiles — 12 MB — a launch is all latency and k_round's 4-wave workgroups spread a small sweep over
        // more compute units than 16-wave ones do: 100 M rows stride 1 %, 977 tiles: 6.8 against 7.8 us.  AQE_SINGLE_LEAN_MIN_TILES moves the line.)
        static const uint64_t min_tiles = [] { const char* e = std::getenv("AQE_SINGLE_LEAN_MIN_TILES"); return e ? std::strtoull(e, nullptr, 10) : 1536ull; }();
        if (R == 1 && !p->host.is_random && !p->host.is_perm && !p->host.has_topup && p->grid > 0 && p->rounds[0].ntiles >= ((q->flags & AQE_Q_FORCE_LEAN) ? 1ull : std::max<uint64_t>(min_tiles, 1))) {
            int rc2 = build_lean_form(p.get(), false, p->single_lean, p->grid, 1, false);
            if (rc2 != AQE_OK) return rc2;
        }
    }
    *out = p.release();
    return AQE_OK;
}

int cached_plan(aqe_ctx* c, const aqe_query* q, aqe_plan** out) {
    for (auto& kv : c->cache)
        if (std::memcmp(&kv.first, q, sizeof(aqe_query)) == 0) {
            for (uint64_t step : {kv.second->view_step_rounds, kv.second->view_step_topup}) {  // the views it reads were just used
                auto it = step ? c->stride_views.find(step) : c->stride_views.end();
                if (it != c->stride_views.end()) it->second.last_use = ++c->view_clock;
            }
            *out = kv.second;
            return AQE_OK;
        }
    aqe_plan* p = nullptr;
    int rc = create_plan(c, q, &p);
    if (rc != AQE_OK) return rc;
    if (c->cache.size() >= 64) { destroy_plan(c->cache.front().second); c->cache.erase(c->cache.begin()); }
    p->cached = true;
    for (uint64_t step : {p->view_step_rounds, p->view_step_topup})
        if (step) c->stride_views[step].cache_refs++;
    c->cache.emplace_back(*q, p);
    *out = p;
    return AQE_OK;
}

namespace {

// The descriptor of one persistent sweep of form F of plan p (what k_sweep_persist takes by value and k_sweep_multi reads
// from its table).  `inline_ok`: the family table may travel in the descriptor (kernel arguments of a single launch).
void fill_form(aqe_plan* p, const SweepForm& F, bool totals_only, double* out_totals, unsigned long long epoch, bool inline_ok, PersistLaunch& a) {
    aqe_ctx* c = p->ctx;
    a = PersistLaunch{};
    a.sw = sweep_common(p, F.d_fams, static_cast<uint32_t>(F.h_fams.size()));
    a.ntiles = F.ntiles;
    for (uint32_t r = 0; r <= F.slots; ++r) a.round_begin[r] = F.round_begin[r];
    for (uint32_t r = 0; r < F.slots; ++r) { a.part_first[r] = F.part_first[r]; a.part_count[r] = F.part_count[r]; }
    for (uint32_t r = 0; r <= F.slots; ++r) { a.step_begin[r] = F.step_begin[r]; a.round_mod[r] = F.round_mod[r]; }
    a.rounds = F.slots;
    a.epoch = epoch;
    a.ctl = p->d_ctl;
    a.partials = F.d_ppart;
    a.state = p->d_state;
    a.fold = fold_params(p, false);
    a.fin = finalize_params(p);
    a.result = p->d_result;
    a.result_seq = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p->d_result) + (reinterpret_cast<const volatile char*>(p->h_seq) - reinterpret_cast<const char*>(p->h_result)));
    a.rehearsal_state = static_cast<QueryState*>(p->d_rehearsal);
    a.rehearsal_result = reinterpret_cast<aqe_result*>(static_cast<char*>(p->d_rehearsal) + sizeof(QueryState));
    a.stamps = c->d_stamps;
    a.finalize_here = 1u;
    a.more_rounds = F.more_rounds;
    a.want_ticks = (p->want_ticks && !totals_only) ? 1u : 0u;
    a.topup_slot = totals_only ? 0u : F.topup_slot;
    a.topup_gate = p->host.has_topup ? 1u : 0u;
    a.totals_only = totals_only ? 1u : 0u;
    a.out_totals = out_totals;
    a.inline_fams = (inline_ok && F.h_fams.size() <= static_cast<size_t>(kPersistInlineFams) && F.ntiles < 0xffffffffull) ? 1u : 0u;
    if (a.inline_fams) {
        std::copy(F.h_fams.begin(), F.h_fams.end(), a.fams);
        for (size_t i = 0; i < static_cast<size_t>(kPersistInlineFams); ++i)
            a.fam_begin[i] = i < F.h_fams.size() ? static_cast<uint32_t>(F.h_fams[i].tile_begin) : 0xffffffffu;
    }
}

// The descriptor of the lean variant L of a form (what k_sweep_lean takes by value).
void fill_lean(aqe_plan* p, const SweepForm& L, bool totals_only, double* out_totals, unsigned long long epoch, LeanLaunch& a) {
    a = LeanLaunch{};
    const SweepCommon sw = sweep_common(p, nullptr, 0);
    a.amount = sw.amount;
    a.runs = L.h_runs;
    a.wide = L.wide ? L.d_wide : nullptr;
    a.nruns = L.nruns;
    a.ntiles = static_cast<uint32_t>(L.ntiles);
    a.tiles_per_wg = L.tiles_per_wg;
    a.has_where = sw.has_where;
    a.wmin = sw.wmin; a.wmax = sw.wmax; a.shift = sw.shift;
    a.partials = L.d_ppart;
    a.counter = p->counter;
    a.epoch = epoch;
    LeanTail& t = a.tail;
    t.rounds = L.slots;
    for (uint32_t r = 0; r <= L.slots; ++r) t.slot_begin[r] = L.slot_begin[r];
    t.out_totals = out_totals;
    t.state = p->d_state;
    t.fold = fold_params(p, false);
    t.fin = finalize_params(p);
    t.result = p->d_result;
    t.result_seq = reinterpret_cast<unsigned long long*>(reinterpret_cast<char*>(p->d_result) + (reinterpret_cast<const volatile char*>(p->h_seq) - reinterpret_cast<const char*>(p->h_result)));
    t.finalize_here = 1u;
    t.topup_gate = p->host.has_topup ? 1u : 0u;
    t.more_rounds = L.more_rounds;
    t.topup_slot = totals_only ? 0u : L.topup_slot;
    t.want_ticks = (p->want_ticks && !totals_only) ? 1u : 0u;
    t.totals_only = totals_only ? 1u : 0u;
    // (aqe_plan_enqueue_all puts the device-gated top-up launch behind this one when the plan's last execution needed it)
    t.keep_state = (!totals_only && p->host.has_topup && p->expect_topup && !L.topup_slot) ? 1u : 0u;
}

int launch_lean(aqe_plan* p, const SweepForm& L, bool totals_only, double* out_totals, hipStream_t s) {
    aqe_ctx* c = p->ctx;
    LeanLaunch a;
    fill_lean(p, L, totals_only, out_totals, c->epoch++, a);
    p->poll_epoch = totals_only ? 0 : a.epoch;
    p->last_exec = totals_only ? 2 : 1;
    p->last_kernel = AQE_KERNEL_SWEEP_LEAN;
    p->last_samples = L.samples;
    p->last_topup_swept = L.topup_slot != 0 && !totals_only;
    p->last_grid = L.grid;
    p->last_first_unswept = L.slots - L.topup_slot;
    const bool prof = p->profile && 2 * (p->lev_used + 1) <= p->lev.size();
    HIPCHK(c, launch_sweep_lean(a, L.grid, p->nt, s, prof ? p->lev[2 * p->lev_used] : nullptr, prof ? p->lev[2 * p->lev_used + 1] : nullptr));
    if (prof) p->lev_used++;
    return AQE_OK;
}

int launch_form(aqe_plan* p, const SweepForm& F, bool totals_only, double* out_totals, hipStream_t s) {
    aqe_ctx* c = p->ctx;
    // the lean kernel when the plan qualifies for it (and nobody asked for the persistent sweep's in-kernel timeline)
    const SweepForm* lean = &F == &p->decide ? &p->decide_lean : &F == &p->head ? &p->head_lean : &F == &p->totals ? &p->totals_lean : nullptr;
    // A sweep that is in flight all at once (two tiles per wave: 64 MB) has nothing to gain from a monitor; a longer
    // one takes the lean launch when the rules are predicted not to hold before the plan runs out of rounds (create_plan)
    // — should the prediction fail, the answer is the same and the rounds behind the stopping one were swept for nothing.
    // (The head form is judged once, at the end, by either kernel.)
    const bool lean_pays = lean && lean->ok && (lean->ntiles <= 2ull * lean->grid * kPersistWaves || p->predicted_full || &F == &p->head);
    if (lean_pays && !c->d_stamps && !(p->q.flags & AQE_Q_NO_LEAN)) return launch_lean(p, *lean, totals_only, out_totals, s);
    {
        int rc = materialize_form(c, const_cast<SweepForm&>(F));  // (the plan's own form: first use)
        if (rc != AQE_OK) return rc;
    }
    PersistLaunch a;
    fill_form(p, F, totals_only, out_totals, c->epoch++, true, a);
    p->poll_epoch = totals_only ? 0 : a.epoch;
    p->last_exec = totals_only ? 2 : 1;
    p->last_kernel = AQE_KERNEL_SWEEP_PERSIST;
    p->last_samples = F.samples;
    p->last_topup_swept = F.topup_slot != 0 && !totals_only;
    p->last_grid = F.grid;
    p->last_first_unswept = F.slots - F.topup_slot;
    if (c->d_stamps) {
        HIPCHK(c, hipMemsetAsync(c->d_stamps, 0, 8 * (8 * static_cast<size_t>(c->persist_grid) * kPersistWaves + 8 * kMaxPersistRounds), s));
        HIPCHK(c, hipStreamSynchronize(s));
    }
    const bool prof = p->profile && 2 * (p->lev_used + 1) <= p->lev.size();
    HIPCHK(c, launch_sweep_persist(a, F.grid, s, prof ? p->lev[2 * p->lev_used] : nullptr, prof ? p->lev[2 * p->lev_used + 1] : nullptr));
    if (prof) p->lev_used++;
    return AQE_OK;
}

}  // namespace

int enqueue_all(aqe_plan* p, hipStream_t s, bool timed) {
    aqe_ctx* c = p->ctx;
    // Queries are timed on the device clock (the first launch notes its start — k_round in the state, the monitor of a
    // persistent launch in LDS — and whoever finishes the query writes the elapsed ticks into the result): no event
    // records in the queue, and the result can be polled.  A replayed graph (frozen arguments) and a profiled plan
    // keep the two event records.
    const bool graph_launch = !(p->persist && (!p->per_round || p->head.ok)) && p->rounds.size() >= kGraphMinRounds &&
                              p->rounds.size() <= kGraphMaxRounds && !p->profile && !std::getenv("AQE_NO_GRAPH");
    const bool tick_timing = timed && !p->profile && !graph_launch && !p->rounds.empty();
    p->want_ticks = tick_timing;
    if (timed && !tick_timing) HIPCHK(c, hipEventRecord(p->ev0, s));  // an event record is a queue packet: off the throughput path
    // (taking the first launch's begin and the last launch's end from the dispatches themselves — hipExtLaunchKernelGGL —
    // was tried instead of the two records: the reported time loses the records' 3 us, the call gains 8 us of host work)
    p->lev_used = 0;
    p->poll_epoch = 0;
    unsigned long long plain_epoch = 0;
    bool topup_done = false;
    if (p->rounds.empty()) {  // nothing to sample (empty table / zero target): a zero state, finalized
        HIPCHK(c, hipMemsetAsync(p->d_state, 0, sizeof(QueryState), s));
        HIPCHK(c, launch_finalize(p->d_state, finalize_params(p), p->d_result, s));
    } else {
        if (p->persist && (!p->per_round || p->head.ok)) {
            const SweepForm& F = p->per_round ? p->head : p->decide;
            int rc = launch_form(p, F, false, nullptr, s);
            if (rc != AQE_OK) return rc;
            // The monitor has written the result.  The top-up (DB.cpp:1031-1040) is rarely due — only when the query
            // stops with fewer than base/4 rows — so its launch is enqueued only for plans whose last execution
            // needed it; otherwise the result carries topup_pending if it was due after all, and fetch() runs it.
            topup_done = p->last_topup_swept || !p->expect_topup;
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
        } else if (p->single_lean.ok && !(p->q.flags & AQE_Q_NO_LEAN) && !c->d_stamps) {  // a single-round plan of runs / blocks (no top-up)
            int rc = launch_lean(p, p->single_lean, false, nullptr, s);
            if (rc != AQE_OK) return rc;
        } else {
            plain_epoch = c->epoch++;  // launch by launch: the one that finishes the query writes the check word
            for (uint32_t i = 0; i < p->rounds.size(); ++i) {
                int rc = enqueue_launch(p, p->rounds[i], i, false, true, nullptr, s, plain_epoch);
                if (rc != AQE_OK) return rc;
            }
            p->poll_epoch = plain_epoch;
        }
        if (p->host.has_topup && !topup_done) {
            int rc = enqueue_launch(p, p->topup, static_cast<uint32_t>(p->rounds.size()), true, true, nullptr, s, plain_epoch);
            if (rc != AQE_OK) return rc;
            p->poll_epoch = plain_epoch;  // the result comes from the top-up launch (behind a persistent launch: no check word)
        }
    }
    if ((timed && !tick_timing) || c->d_stamps) p->poll_epoch = 0;  // event timings and stamps are read after the launch has ended
    if (timed && !tick_timing) HIPCHK(c, hipEventRecord(p->ev1, s));
    p->want_ticks = false;
    p->timed = timed;
    p->tick_timed = tick_timing;
    return AQE_OK;
}

int fetch(aqe_plan* p, aqe_result* out, hipStream_t s, bool already_synced) {
    aqe_ctx* c = p->ctx;
    // A persistent launch writes its result into pinned host memory and a check word beside it (kernels.hpp,
    // result_check): the host reads the result from there as soon as all of it has landed, instead of waiting for the
    // launch to drain and its completion signal to travel (5 us of a 30 us closed loop).  Anything else — and a
    // result that never shows up — takes the stream.
    bool landed = false;
    static const bool no_poll = std::getenv("AQE_NO_POLL") != nullptr;  // diagnostics: always wait for the stream
    if (p->poll_epoch != 0 && !no_poll) {
        const auto t0 = std::chrono::steady_clock::now();
        const volatile unsigned long long* src = reinterpret_cast<const volatile unsigned long long*>(p->h_result);
        static_assert(sizeof(aqe_result) % 8 == 0, "the result is read word by word");
        aqe_result snap;
        for (unsigned spins = 0; !landed; ++spins) {
            unsigned long long w[sizeof(aqe_result) / 8];
            for (size_t i = 0; i < sizeof(aqe_result) / 8; ++i) w[i] = src[i];
            std::memcpy(&snap, w, sizeof snap);
            landed = *p->h_seq == result_check(snap, p->poll_epoch);
            if (!landed && (spins & 255u) == 255u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        landed = landed && snap.topup_pending == 0 && snap.device_status == 0;  // more to launch, or to report: the ordinary way
    }
    p->poll_epoch = 0;  // consumed: only aqe_plan_enqueue_all arms it
    if (!landed && !already_synced) HIPCHK(c, hipStreamSynchronize(s));
    p->want_ticks = p->timed && p->tick_timed;  // launches made from here continue the device-clock timing
    if (p->last_exec == 1 && p->h_result->topup_pending == 2) {
        // the head form ran out of rounds before the query stopped (the prediction failed): the remaining rounds go
        // out one launch each, the top-up behind them — and from now on this plan takes the full single launch
        for (uint32_t i = p->last_first_unswept; i < p->rounds.size(); ++i) {
            int rc = enqueue_launch(p, p->rounds[i], i, false, true, nullptr, s);
            if (rc != AQE_OK) return rc;
        }
        if (p->host.has_topup) {
            int rc = enqueue_launch(p, p->topup, static_cast<uint32_t>(p->rounds.size()), true, true, nullptr, s);
            if (rc != AQE_OK) return rc;
        }
        HIPCHK(c, hipStreamSynchronize(s));
        p->per_round = false;
        p->expect_topup = p->h_result->topup > 0;
    } else if (p->last_exec == 1 && p->host.has_topup) {
        if (p->h_result->topup_pending) {  // the top-up was due and its launch had not been enqueued: run it now
            int rc = enqueue_launch(p, p->topup, static_cast<uint32_t>(p->rounds.size()), true, true, nullptr, s);
            if (rc != AQE_OK) return rc;
            HIPCHK(c, hipStreamSynchronize(s));
        }
        p->expect_topup = p->h_result->topup > 0;  // the next execution enqueues the top-up launch up front, or not
    }
    p->want_ticks = false;
    *out = *p->h_result;
    if (c->d_stamps && p->persist) {
        const size_t W = static_cast<size_t>(p->last_grid) * kPersistWaves;
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
                if (q[3]) std::fprintf(f, " ..r%zu: seen %.2f folded %.2f rules %.2f judged %.2f |", r, us(q[3]), us(q[4]), us(q[7]), us(q[5]));
            }
            std::fprintf(f, "\n");
            std::fclose(f);
        }
    }
    if (out->device_status != 0) {  // the monitor gave up waiting for a workgroup's partial
        return fail(c, AQE_ERR_HIP, "device-side round protocol timed out");
    }
    if (p->timed && !p->tick_timed) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p->ev0, p->ev1) == hipSuccess) out->kernel_ms = ms;
    }
    return AQE_OK;
}

// Synchronous callers (aqe_reduce, aqe_gather).  A plan of very many rounds — the reference's own cadence, ten rows
// per worker and round — would enqueue tens of thousands of launches of which all but the first few are device-side
// no-ops once should_stop is set.  Here the host enqueues a chunk of rounds, looks at should_stop, and stops
// launching when it is set.  (aqe_plan_enqueue_all stays fully asynchronous: it enqueues every round.)
constexpr uint32_t kSyncChunkRounds = 256;

int run_sync(aqe_plan* p, hipStream_t s, bool timed) {
    aqe_ctx* c = p->ctx;
    if ((p->persist && (!p->per_round || p->head.ok)) || p->rounds.size() <= kSyncChunkRounds) return enqueue_all(p, s, timed);
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
    p->tick_timed = false;
    return AQE_OK;
}

namespace {

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

}  // namespace
}  // namespace aqe

// One-launch form of a batch (k_sweep_multi, persist.hip): the grid is cut into one group of workgroups per plan, group i
// sweeps plan i's rounds with a form built for the group's size.  kind 0: decisions taken in the kernel (single GPU:
// the plan's decide form, or its head form when the query is predicted to stop early); kind 1: totals only (multi-GPU).
struct BatchMulti {
    bool built = false;
    std::vector<SweepForm> forms;          // group form of plan i
    PersistLaunch* d_table = nullptr;      // [plans] descriptors, written once
    bool lean = false;                     // every plan qualifies for the lean launch: groups of k_sweep_lean_multi
    LeanLaunch* d_ltable = nullptr;        // ... and their descriptors
    unsigned long long* d_wgmap = nullptr; // [grid] workgroup -> (plan, group size, index in the group)
    unsigned grid = 0;
    uint64_t samples = 0;                  // rows one launch sweeps (all plans)
    bool nt = false;                       // the launch's loads go past the caches (build_multi)
    const double* totals = nullptr;        // kind 1: the buffer the descriptors were written for
    uint64_t stride = 0;
};

struct aqe_batch {
    aqe_ctx* ctx = nullptr;
    std::vector<aqe_plan*> plans;
    std::vector<hipEvent_t> swept;  // lane l's sweeps of this batch are enqueued up to here
    hipEvent_t reduced = nullptr;   // the caller's stream up to (and including) the collective and the replays
    ReplayItem* d_items = nullptr;  // one entry per plan, for the single replay launch
    BatchMulti multi[2];
    hipStream_t last_stream = nullptr;  // where the most recent one-launch execution went
    int last_kind = -1;
    bool profile = false;               // aqe_batch_set_profiling: the launch carries an event pair on its dispatch
    bool profiled = false;              // ... and the most recent launch did
    hipEvent_t pev0 = nullptr, pev1 = nullptr;
};

namespace aqe {
namespace {

void free_multi(BatchMulti& m) {
    for (SweepForm& f : m.forms)
        if (f.d_ppart) (void)hipFree(f.d_ppart);
    m.forms.clear();
    if (m.d_table) (void)hipFree(m.d_table);
    if (m.d_ltable) (void)hipFree(m.d_ltable);
    if (m.d_wgmap) (void)hipFree(m.d_wgmap);
    m.d_ltable = nullptr;
    m.lean = false;
    m.d_table = nullptr;
    m.d_wgmap = nullptr;
    m.built = false;
}

uint64_t head_tiles(const aqe_plan* p) {
    uint64_t t = p->host.has_topup ? p->topup.ntiles : 0;
    for (size_t r = 0; r < p->r_head && r < p->rounds.size(); ++r) t += p->rounds[r].ntiles;
    return t;
}

// Builds the one-launch form of a batch: group sizes in proportion to the plans' tiles (powers of two, the context's
// persistent grid as the budget), one sweep form per plan for its group's size, the descriptor table, the workgroup map.
int build_multi(aqe_batch* b, int kind, double* dev_totals, uint64_t row_stride) {
    aqe_ctx* c = b->ctx;
    BatchMulti& m = b->multi[kind];
    HIPCHK(c, hipDeviceSynchronize());  // a previous launch may still be reading the old table and partial lists
    free_multi(m);
    const size_t n = b->plans.size();
    std::vector<uint64_t> w(n, 0);
    std::vector<char> head(n, 0);
    uint64_t wsum = 0;
    for (size_t i = 0; i < n; ++i) {
        aqe_plan* p = b->plans[i];
        const size_t R = p->rounds.size();
        if (p->host.is_random || p->host.is_perm || R == 0 || R > static_cast<size_t>(kMaxPersistRounds))
            return fail(c, AQE_ERR_UNSUPPORTED, "a plan of the batch has no single-launch form (seeded-random sampler, empty sample or more than 32 rounds)");
        bool every = true;
        for (size_t r = 0; r < R; ++r) every = every && p->rounds[r].ntiles > 0;
        if (kind == 0 && !(c->shard_lo == 0 && c->n_local == c->n_global && every))
            return fail(c, AQE_ERR_UNSUPPORTED, "in-kernel decisions need the whole table in this context and rows in every round; shards take the totals form");
        head[i] = kind == 0 && p->per_round && p->r_head > 0 && p->r_head <= R;
        if (head[i]) w[i] = head_tiles(p);
        else for (size_t r = 0; r < R; ++r) w[i] += p->rounds[r].ntiles;
        w[i] = std::max<uint64_t>(w[i], 1);
        wsum += w[i];
    }
    // Group sizes: powers of two (the cyclic-run arithmetic of a form wants one), in proportion to the plans' tiles, the
    // context's persistent grid (one workgroup per compute unit) as the budget.  Rounding down leaves up to half the
    // budget unused, so the groups with the most tiles per workgroup are doubled while the budget allows.
    const uint64_t budget = std::max(1u, c->persist_grid);
    std::vector<uint32_t> gs(n, 1);
    auto cap_of = [&](size_t i) {  // no more workgroups than one tile per sweeper wave needs
        uint32_t cap = 1;
        while (cap < static_cast<uint32_t>(kMaxPersistGrid) && static_cast<uint64_t>(cap) * kPersistWaves - 1 < w[i]) cap *= 2;
        return cap;
    };
    uint64_t used = 0;
    for (size_t i = 0; i < n; ++i) {
        const double share = static_cast<double>(budget) * static_cast<double>(w[i]) / static_cast<double>(wsum);
        while (2.0 * gs[i] <= share && 2u * gs[i] <= cap_of(i)) gs[i] *= 2;
        used += gs[i];
    }
    for (;;) {
        size_t best = n;
        double best_load = 0.0;
        for (size_t i = 0; i < n; ++i) {
            if (used + gs[i] > budget || 2u * gs[i] > cap_of(i)) continue;
            const double load = static_cast<double>(w[i]) / gs[i];
            if (load > best_load) { best_load = load; best = i; }
        }
        if (best == n) break;
        used += gs[best];
        gs[best] *= 2;
    }
    m.forms.resize(n);
    std::vector<PersistLaunch> table(n);
    std::vector<LeanLaunch> ltable;
    std::vector<unsigned long long> wgmap, monitors;
    const char* layout_env = std::getenv("AQE_MULTI_LAYOUT");
    const bool packed_layout = layout_env && std::strcmp(layout_env, "packed") == 0;
    m.samples = 0;
    {   // Lean groups (lean.hip, k_sweep_lean_multi) when every plan of the batch qualifies: no monitor waves, the last
        // workgroup of a group to arrive finishes its query.  (AQE_MULTI_LEAN=0: the groups of k_sweep_multi.)
        static const bool lean_off = [] { const char* e = std::getenv("AQE_MULTI_LEAN"); return e && e[0] == '0'; }();
        bool all = !lean_off;
        std::vector<SweepForm> lf(n);
        for (size_t i = 0; i < n && all; ++i) {
            aqe_plan* p = b->plans[i];
            if (p->q.flags & AQE_Q_NO_LEAN) { all = false; break; }
            int rc = build_lean_form(p, head[i] != 0, lf[i], gs[i], head[i] ? p->r_head : p->rounds.size());
            if (rc != AQE_OK) { for (SweepForm& f : lf) if (f.d_ppart) (void)hipFree(f.d_ppart); return rc; }
            all = lf[i].ok && !lf[i].wide;  // (the groups of a batch read their run table out of the batch's descriptor table)
        }
        if (all) {
            m.lean = true;
            ltable.resize(n);
            for (size_t i = 0; i < n; ++i) {
                fill_lean(b->plans[i], lf[i], kind == 1, kind == 1 ? dev_totals + i * row_stride : nullptr, 0, ltable[i]);
                ltable[i].tail.want_ticks = 0;
                ltable[i].tail.keep_state = 0;  // (a batch never enqueues a top-up launch up front: fetch() runs the due ones)
                m.samples += lf[i].samples;
            }
            m.forms = std::move(lf);
        } else {
            for (SweepForm& f : lf) if (f.d_ppart) (void)hipFree(f.d_ppart);
        }
    }
    for (size_t i = 0; i < n && !m.lean; ++i) {
        aqe_plan* p = b->plans[i];
        const size_t R = p->rounds.size();
        const uint32_t g = gs[i];
        SweepForm& F = m.forms[i];
        int rc = AQE_OK;
        if (head[i]) {
            // rounds + top-up as one more slot (the monitor judges once, from one window of steps), else the rounds alone
            bool done = false;
            if (p->host.has_topup) {
                rc = build_sweep_form(p, true, F, g, p->r_head);
                if (rc != AQE_OK) return rc;
                if (F.step_begin[F.slots] <= static_cast<uint32_t>(kDecSteps)) done = true;
                else { (void)hipFree(F.d_ppart); F = SweepForm{}; }
            }
            if (!done) rc = build_sweep_form(p, false, F, g, p->r_head);
        } else {
            rc = build_sweep_form(p, false, F, g, R);
        }
        if (rc == AQE_OK) rc = materialize_form(c, F);
        if (rc != AQE_OK) return rc;
        fill_form(p, F, kind == 1, kind == 1 ? dev_totals + i * row_stride : nullptr, 0, false, table[i]);
        table[i].stamps = nullptr;  // (the stamp layout is per launch grid: single launches only)
        table[i].want_ticks = 0;
        m.samples += F.samples;
    }
    // Workgroup order (wg_map): XCD-aware.  Workgroup p runs on compute die p mod 8, each die has its own L2, and
    // workgroup k of a group sweeps the k-th slice of its query's tiles.  A group's workgroups are therefore contiguous,
    // index 0 (the monitor's) first, groups of 8 or more first of all (their blocks then start at multiples of 8):
    // workgroup k of every such group sits on die k mod 8, and queries that sample the same rows — every `--e 0.01` query
    // does: the reference's samplers are deterministic in N and pct — read them out of that die's L2 together (134
    // against 144 us for the bench batch, a fifth of the fabric traffic).  Every die gets its share of every group, and
    // dispatch is in order per die: ahead of anything a monitor waits for there are only sweepers of its own group and
    // workgroups of earlier groups, so a waiting monitor never sits in front of what it waits for.
    // AQE_MULTI_LAYOUT=packed lists every group's sweeper-only workgroups first and all the monitors' workgroups last
    // instead: no alignment, the queries of a batch do not meet in a die's L2, and the measured bandwidth is the memory
    // system's alone (the bench reports that layout beside the default).
    std::vector<size_t> order;
    for (size_t i = 0; i < n; ++i) if (gs[i] >= 8) order.push_back(i);
    for (size_t i = 0; i < n; ++i) if (gs[i] < 8) order.push_back(i);
    for (size_t i : order) {
        const unsigned long long g = gs[i], tag = (static_cast<unsigned long long>(i) << 32) | (g << 16);
        if (packed_layout) monitors.push_back(tag);
        else wgmap.push_back(tag);
        for (unsigned long long k = 1; k < g; ++k) wgmap.push_back(tag | k);
    }
    wgmap.insert(wgmap.end(), monitors.begin(), monitors.end());
    m.grid = static_cast<unsigned>(wgmap.size());
    if (m.lean) {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&m.d_ltable), n * sizeof(LeanLaunch)));
        HIPCHK(c, hipMemcpy(m.d_ltable, ltable.data(), n * sizeof(LeanLaunch), hipMemcpyHostToDevice));
    } else {
        HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&m.d_table), n * sizeof(PersistLaunch)));
        HIPCHK(c, hipMemcpy(m.d_table, table.data(), n * sizeof(PersistLaunch), hipMemcpyHostToDevice));
    }
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&m.d_wgmap), wgmap.size() * sizeof(unsigned long long)));
    HIPCHK(c, hipMemcpy(m.d_wgmap, wgmap.data(), wgmap.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
    {   // Load policy of the launch.  The groups of a batch that sweep the same rows find each other's lines in the caches
        // (default policy); groups over (nearly) disjoint row windows that together exceed the Infinity Cache find
        // nothing again: non-temporal.  Overlap = sum of the plans' row ranges over the length of their union.
        std::vector<std::pair<uint64_t, uint64_t>> iv;
        uint64_t sum_len = 0, swept = 0;
        for (size_t i = 0; i < n; ++i) {
            const aqe_query& q = b->plans[i]->q;
            const uint64_t lo = q.row_hi > q.row_lo ? q.row_lo : 0, hi = q.row_hi > q.row_lo ? q.row_hi : c->n_global;
            iv.emplace_back(lo, hi);
            sum_len += hi - lo;
            swept += m.forms[i].samples;
        }
        std::sort(iv.begin(), iv.end());
        uint64_t uni = 0, cur_lo = 0, cur_hi = 0;
        for (const auto& x : iv) {
            if (x.first > cur_hi) { uni += cur_hi - cur_lo; cur_lo = x.first; cur_hi = x.second; }
            else cur_hi = std::max(cur_hi, x.second);
        }
        uni += cur_hi - cur_lo;
        const double overlap = uni ? static_cast<double>(sum_len) / static_cast<double>(uni) : 1.0;
        m.nt = overlap < 1.5 && static_cast<double>(swept) * sizeof(double) / overlap > static_cast<double>(kInfinityCacheBytes);
    }
    m.totals = dev_totals;
    m.stride = row_stride;
    m.built = true;
    return AQE_OK;
}

int launch_multi(aqe_batch* b, int kind, hipStream_t s) {
    aqe_ctx* c = b->ctx;
    BatchMulti& m = b->multi[kind];
    const unsigned long long epoch = c->epoch++;
    for (size_t i = 0; i < b->plans.size(); ++i) {
        aqe_plan* p = b->plans[i];
        p->poll_epoch = kind == 0 ? epoch : 0;
        p->last_exec = kind == 0 ? 1 : 2;
        p->last_kernel = m.lean ? AQE_KERNEL_SWEEP_LEAN_MULTI : AQE_KERNEL_SWEEP_MULTI;
        p->last_first_unswept = m.forms[i].slots - m.forms[i].topup_slot;
        p->last_grid = m.forms[i].grid;
        p->lev_used = 0;
        p->timed = false;
        p->tick_timed = false;
    }
    b->profiled = b->profile;
    if (m.lean) HIPCHK(c, launch_sweep_lean_multi(m.d_ltable, m.d_wgmap, epoch, m.grid, m.nt, s, b->profile ? b->pev0 : nullptr, b->profile ? b->pev1 : nullptr));
    else HIPCHK(c, launch_sweep_multi(m.d_table, m.d_wgmap, epoch, m.grid, m.nt, s, b->profile ? b->pev0 : nullptr, b->profile ? b->pev1 : nullptr));
    b->last_stream = s;
    b->last_kind = kind;
    return AQE_OK;
}

}  // namespace
}  // namespace aqe

extern "C" {

// ---- plans --------------------------------------------------------------------------------------
int aqe_plan_create(aqe_ctx* c, const aqe_query* q, aqe_plan** out) {
    if (!c || !q || !out) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    return create_plan(c, q, out);
}

int aqe_plan_create_families(aqe_ctx* c, const aqe_query* q, const aqe_family* fams, uint32_t n_fams, uint64_t global_samples, int on_sorted, aqe_plan** out) {
    if (!c || !q || !out || (n_fams && !fams)) return AQE_ERR_INVALID;
    HIPCHK(c, hipSetDevice(c->device));
    aqe_query plain = *q;  // (method only picks the estimators: AQE_M_EXACT = the unscaled sum of the rows given, anything else = a sample's)
    plain.row_lo = plain.row_hi = 0;
    const GivenFamilies given{fams, n_fams, global_samples, on_sorted != 0};
    return create_plan(c, &plain, out, &given);
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

void aqe_batch_destroy(aqe_batch* b) {
    if (!b) return;
    if (b->ctx) {
        (void)hipSetDevice(b->ctx->device);
        for (hipStream_t s : b->ctx->lanes) (void)hipStreamSynchronize(s);
    }
    if (b->ctx) (void)hipDeviceSynchronize();  // fetch() may have returned before the batch's last launch had ended
    for (BatchMulti& m : b->multi) free_multi(m);
    if (b->pev0) (void)hipEventDestroy(b->pev0);
    if (b->pev1) (void)hipEventDestroy(b->pev1);
    for (hipEvent_t e : b->swept) (void)hipEventDestroy(e);
    if (b->reduced) (void)hipEventDestroy(b->reduced);
    if (b->d_items) (void)hipFree(b->d_items);
    delete b;
}

int aqe_batch_create(aqe_plan* const* plans, uint32_t n, aqe_batch** out) {
    if (!plans || !out || n == 0) return AQE_ERR_INVALID;
    for (uint32_t i = 0; i < n; ++i) {
        if (!plans[i] || !plans[i]->ctx || plans[i]->ctx != plans[0]->ctx) return AQE_ERR_INVALID;
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
    std::vector<ReplayItem> items(n);
    for (uint32_t i = 0; i < n; ++i) {
        aqe_plan* p = plans[i];
        items[i] = ReplayItem{static_cast<uint32_t>(p->rounds.size()), p->host.has_topup ? 1u : 0u, fold_params(p, false), finalize_params(p),
                              p->d_state, p->d_result};
    }
    HIPCHK(c, hipMalloc(reinterpret_cast<void**>(&b->d_items), n * sizeof(ReplayItem)));
    HIPCHK(c, hipMemcpy(b->d_items, items.data(), n * sizeof(ReplayItem), hipMemcpyHostToDevice));
    *out = b.release();
    return AQE_OK;
}

int aqe_batch_enqueue_sweeps(aqe_batch* b, double* dev_totals, uint64_t row_stride) {
    if (!b || !dev_totals) return AQE_ERR_INVALID;
    aqe_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    for (aqe_plan* p : b->plans) {
        int rc = plan_is_current(p);
        if (rc != AQE_OK) return rc;
        if (p->rounds.size() < 2) return fail(c, AQE_ERR_UNSUPPORTED, "a plan of the batch has no batched (totals) form (single round: use the per-round calls)");
        if (row_stride < static_cast<uint64_t>(p->rounds.size()) * kVec) return fail(c, AQE_ERR_INVALID, "row_stride shorter than a plan's totals");
    }
    BatchMulti& m = b->multi[1];
    if (!m.built || m.totals != dev_totals || m.stride != row_stride) {
        int rc = build_multi(b, 1, dev_totals, row_stride);
        if (rc != AQE_OK) return rc;
    }
    // ONE launch sweeps every plan's rounds (a group of workgroups per plan) on the first side stream; the plans'
    // previous replay precedes it there (aqe_batch_enqueue_replays makes the side streams wait for it)
    int rc = launch_multi(b, 1, c->lanes[0]);
    if (rc != AQE_OK) return rc;
    HIPCHK(c, hipEventRecord(b->swept[0], c->lanes[0]));  // (one launch on one side stream: one event)
    return AQE_OK;
}

int aqe_batch_enqueue_all(aqe_batch* b, void* stream) {
    if (!b) return AQE_ERR_INVALID;
    aqe_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    BatchMulti& m = b->multi[0];
    bool stale = !m.built;
    for (size_t i = 0; i < b->plans.size(); ++i) {
        aqe_plan* p = b->plans[i];
        int rc = plan_is_current(p);
        if (rc != AQE_OK) return rc;
        // a head form whose prediction failed (fetch continued the query round by round): the plan takes the full form now
        if (m.built && m.forms[i].more_rounds && !p->per_round) stale = true;
    }
    if (stale) {
        int rc = build_multi(b, 0, nullptr, 0);
        if (rc != AQE_OK) return rc;
    }
    int rc = launch_multi(b, 0, s);
    if (rc != AQE_OK) return rc;
    // The reference's top-up (DB.cpp:1031-1040) is rarely due: its launch (device-gated) goes out only for plans whose
    // previous execution needed it and whose form does not sweep it as a slot; otherwise fetch runs it when the result asks.
    for (size_t i = 0; i < b->plans.size(); ++i) {
        aqe_plan* p = b->plans[i];
        if (p->host.has_topup && p->expect_topup && !m.forms[i].topup_slot) {
            rc = enqueue_launch(p, p->topup, static_cast<uint32_t>(p->rounds.size()), true, true, nullptr, s);
            if (rc != AQE_OK) return rc;
        }
    }
    return AQE_OK;
}

int aqe_batch_set_profiling(aqe_batch* b, int enable) {
    if (!b) return AQE_ERR_INVALID;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    if (enable && !b->pev0) {
        HIPCHK(b->ctx, hipEventCreate(&b->pev0));
        HIPCHK(b->ctx, hipEventCreate(&b->pev1));
    }
    b->profile = enable != 0;
    return AQE_OK;
}

int aqe_batch_launch_info(aqe_batch* b, float* ms, uint64_t* samples, uint32_t* workgroups) {
    if (!b) return AQE_ERR_INVALID;
    if (b->last_kind < 0) return fail(b->ctx, AQE_ERR_INVALID, "no one-launch execution yet");
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    const BatchMulti& m = b->multi[b->last_kind];
    if (samples) *samples = m.samples;
    if (workgroups) *workgroups = m.grid;
    if (ms) {
        if (!b->profiled) return fail(b->ctx, AQE_ERR_INVALID, "the most recent launch was not profiled (aqe_batch_set_profiling)");
        HIPCHK(b->ctx, hipEventSynchronize(b->pev1));
        HIPCHK(b->ctx, hipEventElapsedTime(ms, b->pev0, b->pev1));
    }
    return AQE_OK;
}

int aqe_batch_join(aqe_batch* b, void* stream) {
    if (!b) return AQE_ERR_INVALID;
    aqe_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t main_s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    HIPCHK(c, hipStreamWaitEvent(main_s, b->swept[0], 0));
    return AQE_OK;
}

int aqe_batch_enqueue_replays(aqe_batch* b, const double* dev_totals, uint64_t row_stride, void* stream) {
    if (!b || !dev_totals) return AQE_ERR_INVALID;
    aqe_ctx* c = b->ctx;
    HIPCHK(c, hipSetDevice(c->device));
    hipStream_t main_s = stream ? static_cast<hipStream_t>(stream) : c->stream;
    for (aqe_plan* p : b->plans) {
        int rc = plan_is_current(p);
        if (rc != AQE_OK) return rc;
    }
    // ONE launch replays every plan of the batch on the caller's stream, right behind the collective; the side
    // streams then wait for it before they sweep again (their plans' state, result and buffer row are read here)
    HIPCHK(c, launch_replay_batch(b->d_items, static_cast<uint32_t>(b->plans.size()), dev_totals, row_stride, main_s));
    HIPCHK(c, hipEventRecord(b->reduced, main_s));
    HIPCHK(c, hipStreamWaitEvent(c->lanes[0], b->reduced, 0));  // the side stream sweeps again only behind the replays
    return AQE_OK;
}

int aqe_batch_fetch(aqe_batch* b, aqe_result* out_n) {
    if (!b || !out_n) return AQE_ERR_INVALID;
    HIPCHK(b->ctx, hipSetDevice(b->ctx->device));
    // totals form: the replay ran on the caller's stream and the side streams wait for it — one synchronisation per side
    // stream serves every plan (a synchronisation per plan was most of the host's time per step); one-launch form: each
    // result is polled out of its pinned block, the launch's stream is the fallback
    const bool totals = b->last_kind != 0;
    // (the event behind this batch's replays — not the side stream: that one may already hold the NEXT batch's sweeps,
    // and waiting for those would keep the host from enqueueing the step after)
    if (totals) HIPCHK(b->ctx, hipEventSynchronize(b->reduced));
    for (size_t i = 0; i < b->plans.size(); ++i) {
        int rc = plan_is_current(b->plans[i]);
        hipStream_t s = totals ? b->ctx->lanes[0] : b->last_stream;
        if (rc == AQE_OK) rc = fetch(b->plans[i], out_n + i, s, totals);
        if (rc != AQE_OK) return rc;
    }
    return AQE_OK;
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
    if (p->tick_timed) {  // timed by the device clock: the figure is in the fetched result
        *ms = static_cast<float>(p->h_result->kernel_ms);
        return AQE_OK;
    }
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

int aqe_plan_last_kernel(const aqe_plan* p, int* kernel) {
    if (!p || !kernel) return AQE_ERR_INVALID;
    *kernel = p->last_kernel;
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
    const bool with_topup = p->host.has_topup && (form != 1 || p->expect_topup);  // the single-launch form enqueues it on demand
    const uint32_t n = sweeps + (with_topup ? 1u : 0u);
    *n_out = n;
    if (!samples) return AQE_OK;
    if (cap < n) return AQE_ERR_CAPACITY;
    if (form == 1) samples[0] = p->last_exec == 1 ? p->last_samples : p->decide.samples;
    else for (size_t i = 0; i < p->rounds.size(); ++i) samples[i] = p->rounds[i].samples;
    if (with_topup) samples[sweeps] = p->topup.samples;
    return AQE_OK;
}

}  // extern "C"
