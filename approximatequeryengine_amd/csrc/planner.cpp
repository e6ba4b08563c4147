// planner.cpp — host-side planning of the reference's samplers as arithmetic row families.
//
// Each builder re-derives the index arithmetic of one reference sampler (cited per function; DB.cpp =
// /root/reference/src/aqe_backend/core/custom_bplus_db.cpp) and emits it in closed form:
//     row(o) = row0 + (o / seg_len) * pitch + (o % seg_len) * step,   o in [ord_lo, ord_hi)
// so a GPU can enumerate the sample without materialising it.  Truncating casts (int target counts,
// left-to-right double products) are kept exactly as the reference has them: they decide the sample.
#include "planner.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <regex>

namespace aqe {
namespace {

using u64 = uint64_t;
using u128 = unsigned __int128;

inline int target_of(u64 rows, double pct) { return static_cast<int>(static_cast<double>(rows) * pct / 100.0); }
inline u64 ceil_div(u64 a, u64 b) { return (a + b - 1) / b; }

// number of terms of start, start+step, ... that are < end
inline u64 prog_count(u64 start, u64 end, u64 step) { return start < end ? ceil_div(end - start, step) : 0; }

aqe_family strided(u64 row0, u64 step, u64 count, uint32_t group = 0) {
    aqe_family f{};
    f.row0 = row0;
    f.step = step;
    f.seg_len = count ? count : 1;
    f.pitch = 0;
    f.ord_lo = 0;
    f.ord_hi = count;
    f.group = group;
    return f;
}

aqe_family blocks(u64 row0, u64 pitch, u64 seg_len, u64 total) {
    aqe_family f{};
    f.row0 = row0;
    f.pitch = pitch;
    f.seg_len = seg_len;
    f.step = 1;
    f.ord_lo = 0;
    f.ord_hi = total;
    return f;
}

// Rows of a family ascend with the ordinal (segments never overlap), so the ordinals whose rows fall
// in [lo, hi) form one window; find it by bisection on the closed form.
u64 first_ordinal_at_or_after(const aqe_family& f, u64 a, u64 b, u64 row) {
    while (a < b) {
        u64 mid = a + (b - a) / 2;
        if (family_row(f, mid) >= row) b = mid; else a = mid + 1;
    }
    return a;
}

void clip_push(std::vector<aqe_family>& out, aqe_family f, ClipWindow w) {
    u64 a = 0, b = 0;
    if (f.ord_hi > f.ord_lo) {
        a = first_ordinal_at_or_after(f, f.ord_lo, f.ord_hi, w.lo);
        b = first_ordinal_at_or_after(f, a, f.ord_hi, w.hi);
    }
    f.ord_lo = a;
    f.ord_hi = b > a ? b : a;
    if (f.flags & AQE_F_PAIR) {  // second pointer: row_b(o) = row0_b + o*step, single segment
        aqe_family g = f;
        g.row0 = f.row0_b; g.ord_lo = f.ord_lo_b; g.ord_hi = f.ord_hi_b; g.pitch = 0;
        g.seg_len = ~0ull;  // one segment: o / seg_len == 0
        u64 c = 0, d = 0;
        if (g.ord_hi > g.ord_lo) {
            c = first_ordinal_at_or_after(g, g.ord_lo, g.ord_hi, w.lo);
            d = first_ordinal_at_or_after(g, c, g.ord_hi, w.hi);
        }
        f.ord_lo_b = c;
        f.ord_hi_b = d > c ? d : c;
        if (f.ord_hi_b <= f.ord_lo_b) {  // only the first pointer has rows here: plain family
            f.flags &= ~AQE_F_PAIR;
            f.row0_b = f.ord_lo_b = f.ord_hi_b = 0;
        } else if (f.ord_hi <= f.ord_lo) {  // only the second pointer: make it the plain family
            f.row0 = f.row0_b; f.ord_lo = f.ord_lo_b; f.ord_hi = f.ord_hi_b; f.group = 1;
            f.flags &= ~AQE_F_PAIR;
            f.row0_b = f.ord_lo_b = f.ord_hi_b = 0;
        }
    }
    if (f.ord_hi <= f.ord_lo) return;
    out.push_back(f);
}

// ---- MT19937 + libstdc++-11 uniform_int_distribution (bits/uniform_int_dist.h:246-317) ----------
struct Mt19937 {
    uint32_t mt[624];
    int idx;
    explicit Mt19937(uint32_t seed) {
        mt[0] = seed;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + static_cast<uint32_t>(i);
        idx = 624;
    }
    uint32_t next() {
        if (idx >= 624) {
            for (int k = 0; k < 624; ++k) {
                uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
                mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
            }
            idx = 0;
        }
        uint32_t y = mt[idx++];
        y ^= y >> 11;
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= y >> 18;
        return y;
    }
};

inline uint32_t lemire_bounded(Mt19937& g, uint32_t range) {
    u64 product = static_cast<u64>(g.next()) * range;
    uint32_t low = static_cast<uint32_t>(product);
    if (low < range) {
        uint32_t threshold = static_cast<uint32_t>(0u - range) % range;
        while (low < threshold) {
            product = static_cast<u64>(g.next()) * range;
            low = static_cast<uint32_t>(product);
        }
    }
    return static_cast<uint32_t>(product >> 32);
}

inline u64 splitmix64_at(u64 seed, u64 i) {
    u64 z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

#define AQE_FAIL(msg) do { err = (msg); return AQE_ERR_INVALID; } while (0)

}  // namespace

void clip_family(std::vector<aqe_family>& out, const aqe_family& f, ClipWindow w) { clip_push(out, f, w); }

int random_pointer_indices(u64 N, double pct, uint32_t seed, ClipWindow shard, std::vector<u64>& out,
                           std::string& err) {
    out.clear();
    if (N == 0) return AQE_OK;
    int target = target_of(N, pct);
    if (target <= 0) return AQE_OK;
    if (N > 0xFFFFFFFFull) AQE_FAIL("random_pointer_sample: tables above 2^32-1 rows are not supported");
    u64 want = std::min<u64>(static_cast<u64>(target), N);
    std::vector<u64> bits((N + 63) / 64, 0);
    Mt19937 rng(seed);
    u64 have = 0;
    while (have < want) {  // std::set insertion until `target` unique positions, DB.cpp:872-875
        u64 v = lemire_bounded(rng, static_cast<uint32_t>(N));
        u64 m = 1ull << (v & 63);
        if (!(bits[v >> 6] & m)) { bits[v >> 6] |= m; ++have; }
    }
    u64 lo = std::min(shard.lo, N), hi = std::min(shard.hi, N);
    for (u64 w = lo / 64; w < (hi + 63) / 64; ++w) {  // ascending, as the std::set iterates
        u64 word = bits[w];
        while (word) {
            u64 i = w * 64 + static_cast<u64>(__builtin_ctzll(word));
            word &= word - 1;
            if (i >= lo && i < hi) out.push_back(i);
        }
    }
    return AQE_OK;
}

// ---- the CLI's samplers for small tables (enhanced_aqe_cli.py:181-186): explicit index lists, like random_pointer_sample ----

// Leaf sizes of the reference's B+ tree after N ascending inserts (what insert_batch — it sorts by id, DB.cpp:196-208 —
// and load_from_file produce): a leaf splits when it reaches MAX_KEYS = 255 keys, keeps MAX_KEYS / 2 = 127 and hands 128
// to the new right sibling (DB.cpp:43-62, 211-221), which alone grows from then on.  So every leaf but the last holds 127
// rows and the last one 128 ... 254 (a table below 255 rows is one leaf).
void reference_leaves(u64 N, u64& full_leaves, u64& last_size) {
    if (N < 255) { full_leaves = 0; last_size = N; return; }
    full_leaves = (N - 255) / 127 + 1;
    last_size = N - 127 * full_leaves;
}

// direct_access_sample (DB.cpp:584-644): ~10 % of the leaves at a fixed node step, evenly spaced records inside each.
// Rows in the reference's order (ascending leaves; a leaf visited twice — more nodes asked for than the tree has — gives
// its rows twice: duplicates are part of the reference's sample).
int direct_access_indices(u64 N, double pct, ClipWindow shard, std::vector<u64>& out, u64& global_samples, std::string& err) {
    out.clear();
    global_samples = 0;
    if (N == 0 || !(pct > 0.0)) return AQE_OK;
    auto emit = [&](u64 row) { ++global_samples; if (row >= shard.lo && row < shard.hi) out.push_back(row); };
    if (pct >= 100.0) { for (u64 i = 0; i < N; ++i) emit(i); return AQE_OK; }
    const u64 target = static_cast<u64>(static_cast<double>(N) * pct / 100.0);  // size_t(total * pct / 100.0)
    u64 full = 0, last = 0;
    reference_leaves(N, full, last);
    const u64 L = full + 1;
    if (L > 0x7fffffffull) AQE_FAIL("direct_access_sample: table too large");
    const u64 nodes_to_sample = std::max<u64>(1, target / 10);
    const double node_step = static_cast<double>(L) / static_cast<double>(nodes_to_sample);
    u64 count = 0;
    for (u64 i = 0; i < nodes_to_sample && count < target; ++i) {
        const u64 node = static_cast<u64>(static_cast<double>(i) * node_step);
        if (node >= L) continue;
        const int kc = static_cast<int>(node < full ? 127 : last);
        const u64 start = node < full ? node * 127 : full * 127;
        int rpn = std::max(1, static_cast<int>(target / nodes_to_sample));
        rpn = std::min(rpn, kc);
        const double record_step = static_cast<double>(kc) / rpn;
        for (int j = 0; j < rpn && count < target; ++j) {
            const int ri = static_cast<int>(j * record_step);
            if (ri < kc) { emit(start + static_cast<u64>(ri)); ++count; }
        }
    }
    return AQE_OK;
}

// optimized_sequential_sample (DB.cpp:366-428): systematic, one row whenever the running count reaches the next sample
// point, which advances by step = 100 / pct (a double, accumulated as the reference accumulates it) from a start in
// [0, step).  The reference draws the start from std::random_device; here it comes from the query's seed the way
// libstdc++ would draw it from mt19937(seed): uniform_real_distribution<double>(0, step) = generate_canonical<double, 53>
// (two 32-bit draws, low word first) * step.
int optimized_sequential_indices(u64 N, double pct, uint32_t seed, ClipWindow shard, std::vector<u64>& out, u64& global_samples, std::string& err) {
    (void)err;
    out.clear();
    global_samples = 0;
    if (N == 0 || !(pct > 0.0)) return AQE_OK;
    auto emit = [&](u64 row) { ++global_samples; if (row >= shard.lo && row < shard.hi) out.push_back(row); };
    if (pct >= 100.0) { for (u64 i = 0; i < N; ++i) emit(i); return AQE_OK; }
    const u64 target = static_cast<u64>(static_cast<double>(N) * pct / 100.0);
    if (target == 0) return AQE_OK;
    const double step = 100.0 / pct;
    Mt19937 g(seed);
    const double lo32 = static_cast<double>(g.next()), hi32 = static_cast<double>(g.next());
    double canon = (lo32 + hi32 * 4294967296.0) / 18446744073709551616.0;
    if (canon >= 1.0) canon = std::nextafter(1.0, 0.0);
    double next = canon * step + 0.0;  // (b - a) * canonical + a
    u64 taken = 0, count = 0;
    while (taken < target) {
        // the first running count (1-based) with count >= next: rows are visited one by one, a row is taken at most once
        u64 c = next <= 1.0 ? 1 : static_cast<u64>(std::ceil(next));
        if (c <= count) c = count + 1;
        if (c > N) break;
        count = c;
        emit(c - 1);
        ++taken;
        next += step;
    }
    return AQE_OK;
}

static int build_plan_whole(const aqe_query& q, u64 N, ClipWindow shard, HostPlan& P, std::string& err, const double* zone_var);

// Row window (key-range pruning): plan over the window as if it were the whole table, then shift.
int build_plan(const aqe_query& q, u64 N, ClipWindow shard, HostPlan& P, std::string& err, const double* zone_var) {
    if (q.row_hi <= q.row_lo) return build_plan_whole(q, N, shard, P, err, zone_var);
    if (q.method == AQE_M_ADAPTIVE_BLOCK || q.method == AQE_M_STRATIFIED_BLOCK)
        AQE_FAIL("adaptive/stratified block samplers work on the whole table (no row window)");
    if (q.row_hi > N) AQE_FAIL("row window exceeds the table");
    const u64 lo = q.row_lo, n = q.row_hi - q.row_lo;
    auto rebase = [&](u64 x) { return x <= lo ? 0 : std::min(x - lo, n); };
    aqe_query inner = q;
    inner.row_lo = inner.row_hi = 0;
    int rc = build_plan_whole(inner, n, ClipWindow{rebase(shard.lo), rebase(shard.hi)}, P, err, nullptr);
    if (rc != AQE_OK) return rc;
    for (auto& rf : P.round_fams) for (auto& f : rf) { f.row0 += lo; if (f.flags & AQE_F_PAIR) f.row0_b += lo; }
    for (auto& f : P.topup_fams) f.row0 += lo;
    for (auto& i : P.random_idx) i += lo;
    P.perm_lo = lo;
    return AQE_OK;
}

static int build_plan_whole(const aqe_query& q, u64 N, ClipWindow shard, HostPlan& P, std::string& err, const double* zone_var) {
    P = HostPlan{};
    P.pct = q.sample_percent;
    shard.lo = std::min(shard.lo, N);
    shard.hi = std::min(std::max(shard.hi, shard.lo), N);
    const double pct = q.sample_percent;
    if (!(pct == pct)) AQE_FAIL("sample_percent is NaN");
    const u64 M = (q.visible_rows && q.visible_rows < N) ? q.visible_rows : N;
    P.visible_rows = M;
    const int T = q.num_threads;
    std::vector<aqe_family> fams;  // unclipped families of a single-round sampler
    auto finish_single = [&]() {
        P.rounds = 1;
        P.round_fams.assign(1, {});
        for (const auto& f : fams) {
            // (counted inside the table: at pct > 100 the block samplers plan blocks past the last row, which the reference
            // walks over without taking anything, DB.cpp:1254-1256)
            std::vector<aqe_family> inside;
            clip_push(inside, f, ClipWindow{0, N});
            for (const auto& g : inside) P.global_samples += family_size(g);
            clip_push(P.round_fams[0], f, shard);
        }
        return AQE_OK;
    };

    switch (q.method) {
        case AQE_M_EXACT: {  // DB.cpp:242-274: every row
            if (N) fams.push_back(blocks(0, N, N, N));
            return finish_single();
        }
        case AQE_M_MEMORY_STRIDE: {  // DB.cpp:1540-1566
            if (M == 0) return finish_single();
            int target = target_of(M, pct);
            if (target <= 0) return finish_single();
            u64 stride = q.stride_bytes == 0 ? std::max<u64>(1, M / static_cast<u64>(target))
                                             : std::max<u64>(1, q.stride_bytes / sizeof(aqe_record));
            fams.push_back(strided(0, stride, std::min<u64>(static_cast<u64>(target), ceil_div(M, stride))));
            return finish_single();
        }
        case AQE_M_RANDOM_START_STRIDE: {  // DB.cpp:1838-1878: the stride sampler from a random row in [0, stride)
            if (M == 0) return finish_single();
            int target = target_of(M, pct);
            if (target <= 0) return finish_single();
            u64 stride = q.stride_bytes == 0 ? std::max<u64>(1, M / static_cast<u64>(target))
                                             : std::max<u64>(1, q.stride_bytes / sizeof(aqe_record));
            u64 start = splitmix64_at(q.seed, 0) % stride;  // uniform_int_distribution(0, stride-1), seeded
            fams.push_back(strided(start, stride, std::min<u64>(static_cast<u64>(target), prog_count(start, M, stride))));
            return finish_single();
        }
        case AQE_M_ROWID_MOD: {  // executor.cpp:21-26, 36-41: rowid % step = 0 with rowid = row + 1, step = 100 / int(pct)
            const int ipct = static_cast<int>(pct);
            const u64 step = (ipct <= 0 || ipct >= 100) ? 1 : static_cast<u64>(100 / ipct);
            if (N >= step) fams.push_back(strided(step - 1, step, N / step));
            return finish_single();
        }
        case AQE_M_ADDRESS_ARITHMETIC: {  // DB.cpp:1667-1703
            if (M == 0) return finish_single();
            int target = target_of(M, pct);
            if (target <= 0) return finish_single();
            u64 stride = std::max<u64>(1, M / static_cast<u64>(target));
            fams.push_back(strided(0, stride, std::min<u64>(static_cast<u64>(target), ceil_div(M, stride))));
            return finish_single();
        }
        case AQE_M_FAST_POINTER:    // DB.cpp:737-758
        case AQE_M_SLOW_POINTER: {  // DB.cpp:760-780
            if (N == 0) return finish_single();
            int target = target_of(N, pct);
            if (target <= 0) return finish_single();
            int step = std::max(1, static_cast<int>(N / static_cast<u64>(target)));
            if (q.method == AQE_M_FAST_POINTER) {  // (`step *= step_size` in int, DB.cpp:750: a product past INT_MAX is undefined there — refused here)
                const long long wide = static_cast<long long>(step) * static_cast<long long>(q.step_size);
                if (wide > 0x7fffffffLL) AQE_FAIL("fast_pointer_sample: step * step_size overflows the reference's int");
                step = wide <= 0 ? 0 : static_cast<int>(wide);
            }
            if (step <= 0) AQE_FAIL("fast_pointer_sample: step_size must be positive");
            fams.push_back(strided(0, static_cast<u64>(step),
                                   std::min<u64>(static_cast<u64>(target), ceil_div(N, static_cast<u64>(step)))));
            return finish_single();
        }
        case AQE_M_DUAL_POINTER: {  // DB.cpp:782-813
            if (N == 0) return finish_single();
            int target = target_of(N, pct);
            if (target <= 0) return finish_single();
            int fast_target = target / 3, slow_target = target - fast_target;
            if (fast_target == 0) AQE_FAIL("dual_pointer_sample: target < 3 (the reference divides by zero)");
            const long long fast_wide = 3LL * std::max(1, static_cast<int>(N / static_cast<u64>(fast_target)));  // (`fast_step *= 3` in int, DB.cpp:800)
            if (fast_wide > 0x7fffffffLL) AQE_FAIL("dual_pointer_sample: fast_step * 3 overflows the reference's int");
            u64 fast_step = static_cast<u64>(fast_wide);
            u64 n_fast = std::min<u64>(static_cast<u64>(fast_target), ceil_div(N, fast_step));
            fams.push_back(strided(0, fast_step, n_fast));
            u64 slow_step = static_cast<u64>(std::max(1, static_cast<int>(N / static_cast<u64>(slow_target))));
            u64 offset = fast_step / 2;
            u64 n_slow = std::min<u64>(static_cast<u64>(target) - n_fast, prog_count(offset, N, slow_step));
            fams.push_back(strided(offset, slow_step, n_slow));
            // the two pointers interleave in row order: keep them as separate families (each ascending)
            return finish_single();
        }
        case AQE_M_PARALLEL_POINTER: {  // DB.cpp:815-854
            if (N == 0) return finish_single();
            int target = target_of(N, pct);
            if (target <= 0) return finish_single();
            if (T <= 0) AQE_FAIL("parallel_pointer_sample: num_threads must be positive");
            u64 per_thread = static_cast<u64>(target / T);
            u64 step = static_cast<u64>(std::max(1, static_cast<int>(N / static_cast<u64>(target))));
            for (int t = 0; t < T; ++t) {
                u64 start = (N / static_cast<u64>(T)) * static_cast<u64>(t);
                fams.push_back(strided(start, step, std::min(per_thread, prog_count(start, N, step))));
            }
            return finish_single();
        }
        case AQE_M_BLOCK:   // DB.cpp:1151-1181
        case AQE_M_PAGE: {  // DB.cpp:1183-1216
            if (N == 0) return finish_single();
            int target = target_of(N, pct);
            if (target <= 0) return finish_single();
            u64 B = q.block_size;
            if (q.method == AQE_M_PAGE) { B = q.block_size / sizeof(aqe_record); if (B == 0) B = 1; }
            if (B == 0) AQE_FAIL("block_sample: block_size must be positive");
            u64 total_blocks = ceil_div(N, B);
            u64 to_sample = std::max<u64>(1, static_cast<u64>(static_cast<double>(total_blocks) * pct / 100.0));
            u64 interval = std::max<u64>(1, total_blocks / to_sample);
            u64 nseg = ceil_div(total_blocks, interval);          // blk = 0, interval, ... < total_blocks
            u64 last_row0 = (nseg - 1) * interval * B;
            u64 last_len = std::min(B, N - last_row0);            // only the table's tail block is short
            u64 avail = (nseg - 1) * B + last_len;
            fams.push_back(blocks(0, interval * B, B, std::min<u64>(static_cast<u64>(target), avail)));
            return finish_single();
        }
        case AQE_M_ADAPTIVE_BLOCK: {  // DB.cpp:1273-1329
            if (N == 0) return finish_single();
            int target = target_of(N, pct);
            if (target <= 0) return finish_single();
            const u64 min_b = q.block_size, max_b = q.block_size_max;
            const u64 zones = 10, zone_size = N / zones;
            if (zone_size == 0 || min_b == 0 || max_b < min_b) AQE_FAIL("adaptive_block_sample: needs N >= 10 and 0 < min_block_size <= max_block_size");
            if (!zone_var) AQE_FAIL("adaptive_block_sample: zone variances missing (planned by aqe_reduce after its device pre-pass, or aqe_plan_adaptive_families)");
            double vmax = zone_var[0];
            for (u64 z = 1; z < zones; ++z) vmax = std::max(vmax, zone_var[z]);
            if (!(vmax > 0.0)) AQE_FAIL("adaptive_block_sample: all zone variances are zero (the reference divides 0/0)");
            u64 budget = static_cast<u64>(target);
            for (u64 z = 0; z < zones && budget; ++z) {
                const u64 a = z * zone_size, b = std::min(a + zone_size, N);
                const double ratio = zone_var[z] / vmax;
                const u64 abs_ = min_b + static_cast<u64>(static_cast<double>(max_b - min_b) * (1.0 - ratio));
                const u64 nblk = ceil_div(b - a, abs_), nfull = (b - a) / abs_;
                if (nfull) {  // whole blocks: the first max(1, abs*pct/100) rows of each
                    const u64 cnt = std::min(abs_, std::max<u64>(1, static_cast<u64>(static_cast<double>(abs_) * pct / 100.0)));
                    const u64 total = std::min(budget, nfull * cnt);
                    fams.push_back(blocks(a, abs_, cnt, total));
                    budget -= total;
                }
                if (nblk > nfull && budget) {  // the zone's short last block
                    const u64 i = a + nfull * abs_, len = b - i;
                    const u64 cnt = std::min(len, std::max<u64>(1, static_cast<u64>(static_cast<double>(len) * pct / 100.0)));
                    const u64 total = std::min(budget, cnt);
                    fams.push_back(blocks(i, len, cnt, total));
                    budget -= total;
                }
            }
            return finish_single();
        }
        case AQE_M_STRATIFIED_BLOCK: {  // DB.cpp:1331-1379; rows are positions in the amount-sorted table
            P.on_sorted = true;
            if (N == 0) return finish_single();
            int target = target_of(N, pct);
            if (target <= 0) return finish_single();
            const u64 B = q.block_size;
            const int strata = T;
            if (B == 0 || strata <= 0) AQE_FAIL("stratified_block_sample: block_size and strata_count must be positive");
            const u64 stratum_size = N / static_cast<u64>(strata), per_stratum = static_cast<u64>(target / strata);
            u64 budget = static_cast<u64>(target);
            for (int st = 0; st < strata && budget; ++st) {
                const u64 a = static_cast<u64>(st) * stratum_size, b = (st == strata - 1) ? N : a + stratum_size;
                const u64 recs = b - a, nblocks = ceil_div(recs, B);
                if (nblocks == 0 || per_stratum == 0) continue;  // remaining_samples == 0: nothing is taken
                const u64 to_sample = std::max<u64>(1, static_cast<u64>(static_cast<double>(nblocks) * pct / 100.0));
                const u64 interval = std::max<u64>(1, nblocks / to_sample);
                const u64 nsel = ceil_div(nblocks, interval);
                const u64 take_full = std::min(per_stratum, B);
                const u64 last_len = std::min(B, recs - (nsel - 1) * interval * B);
                const u64 avail = (nsel - 1) * take_full + std::min(per_stratum, last_len);
                const u64 total = std::min(budget, avail);
                fams.push_back(blocks(a, interval * B, take_full, total));
                budget -= total;
            }
            return finish_single();
        }
        case AQE_M_PARALLEL_BLOCK: {  // DB.cpp:1218-1271
            if (N == 0) return finish_single();
            int target = target_of(N, pct);
            if (target <= 0) return finish_single();
            u64 B = q.block_size;
            if (B == 0 || T <= 0) AQE_FAIL("parallel_block_sample: block_size and num_threads must be positive");
            u64 total_blocks = ceil_div(N, B);
            u64 to_sample = std::max<u64>(1, static_cast<u64>(static_cast<double>(total_blocks) * pct / 100.0));
            u64 per_thread = std::max<u64>(1, to_sample / static_cast<u64>(T));
            u64 interval = std::max<u64>(1, total_blocks / to_sample);
            u64 thread_target = static_cast<u64>(target / T);
            for (int t = 0; t < T; ++t) {
                u64 b0 = static_cast<u64>(t) * per_thread, b1 = std::min(b0 + per_thread, to_sample);
                if (b1 <= b0 || thread_target == 0) continue;
                u64 nseg = b1 - b0;
                u64 last_row0 = (b1 - 1) * interval * B;
                u64 last_len = last_row0 < N ? std::min(B, N - last_row0) : 0;
                u64 avail = (nseg - 1) * B + last_len;
                fams.push_back(blocks(b0 * interval * B, interval * B, B, std::min(thread_target, avail)));
            }
            return finish_single();
        }
        case AQE_M_OPTIMIZED_CLT: {  // DB.cpp:1046-1147 (deterministic: both CLT branches return the same rows)
            if (N == 0) return finish_single();
            u64 target = static_cast<u64>(static_cast<double>(N) * pct / 100.0);
            if (target == 0) return finish_single();
            int opt = std::min(T, std::max(1, static_cast<int>(target / 100)));
            if (N < 5000 || target < 200 || opt == 1) {
                u64 step = std::max<u64>(1, N / target);
                fams.push_back(strided(0, step, std::min(target, ceil_div(N, step))));
                return finish_single();
            }
            if (opt <= 0) return finish_single();
            u64 per_thread = target / static_cast<u64>(opt);
            for (int t = 0; t < opt; ++t) {
                u64 rpt = N / static_cast<u64>(opt);
                u64 a = static_cast<u64>(t) * rpt, b = (t == opt - 1) ? N : static_cast<u64>(t + 1) * rpt;
                u64 local = (t == opt - 1) ? target - static_cast<u64>(opt - 1) * per_thread : per_thread;
                if (local == 0) continue;
                u64 stride = std::max<u64>(1, (b - a) / local);
                fams.push_back(strided(a, stride, std::min(local, prog_count(a, b, stride))));
            }
            return finish_single();
        }
        case AQE_M_REGION_STRIDE: {  // DB.cpp:1880-2048 with a proper prefix partition and seeded starts
            if (M == 0) return finish_single();
            if (T <= 0) AQE_FAIL("region stride: num_threads must be positive");
            double per_thread_pct = pct / static_cast<double>(T);
            for (int t = 0; t < T; ++t) {
                u64 a = static_cast<u64>((static_cast<u128>(M) * static_cast<u64>(t)) / static_cast<u64>(T));
                u64 b = static_cast<u64>((static_cast<u128>(M) * static_cast<u64>(t + 1)) / static_cast<u64>(T));
                u64 total = b - a;
                u64 target = static_cast<u64>(static_cast<double>(total) * per_thread_pct / 100.0);
                if (target == 0) continue;
                u64 span = std::min<u64>(total / 10, 100);
                u64 start = a + splitmix64_at(q.seed, static_cast<u64>(t)) % (span + 1);
                u64 stride = std::max<u64>(1, total / target);
                fams.push_back(strided(start, stride, std::min(target, prog_count(start, b, stride))));
            }
            return finish_single();
        }
        case AQE_M_RANDOM_POINTER: {
            P.is_random = true;
            P.rounds = 1;
            int rc = random_pointer_indices(N, pct, static_cast<uint32_t>(q.seed), shard, P.random_idx, err);
            if (rc != AQE_OK) return rc;
            int target = N ? target_of(N, pct) : 0;
            P.global_samples = target > 0 ? std::min<u64>(static_cast<u64>(target), N) : 0;
            return AQE_OK;
        }
        case AQE_M_DIRECT_ACCESS: {
            P.is_random = true;  // (an explicit row list: the kernels of the seeded-random sampler)
            P.rounds = 1;
            return direct_access_indices(N, pct, shard, P.random_idx, P.global_samples, err);
        }
        case AQE_M_OPTIMIZED_SEQUENTIAL: {
            P.is_random = true;
            P.rounds = 1;
            return optimized_sequential_indices(N, pct, static_cast<uint32_t>(q.seed), shard, P.random_idx, P.global_samples, err);
        }
        case AQE_M_RANDOM_DEVICE: {  // sample_records' semantics (DB.cpp:345-363: a uniform prefix of a shuffle), drawn on the device
            P.is_perm = true;
            P.rounds = 1;
            int target = N ? target_of(N, pct) : 0;
            P.perm_n = N;
            P.perm_lo = 0;
            P.perm_target = target > 0 ? std::min<u64>(static_cast<u64>(target), N) : 0;
            P.perm_seed = q.seed;
            P.global_samples = P.perm_target;
            return AQE_OK;
        }
        case AQE_M_CLT_DUAL_POINTER: {  // DB.cpp:885-1043, round-synchronous
            P.is_clt = true;
            P.clt.base = N ? target_of(N, pct) : 0;
            P.clt.z = q.confidence_level >= 0.99 ? 2.576 : q.confidence_level >= 0.95 ? 1.96 : 1.645;
            P.clt.e = q.max_error_percent;
            P.rounds = 0;
            if (q.has_where) AQE_FAIL("the CLT sampler has no WHERE form in the reference");
            if (N == 0 || P.clt.base <= 0) return AQE_OK;  // empty result, DB.cpp:894-898
            if (T <= 0 || T > 4096) AQE_FAIL("clt: num_threads out of range");
            const int F = T / 2, S = T - F;
            if (q.check_interval / 2 == 0) AQE_FAIL("clt: check_interval < 2 (the reference takes a modulo by zero, DB.cpp:993)");
            if (F > 0 && P.clt.base / F == 0) AQE_FAIL("clt: base/fast_threads == 0 (the reference divides by zero, DB.cpp:927)");
            if (S > 0 && P.clt.base / S == 0) AQE_FAIL("clt: base/slow_threads == 0 (the reference divides by zero, DB.cpp:985)");
            P.clt.n_workers = T;
            P.clt.n_fast = F;
            struct W { u64 first, step, count; uint32_t group; };
            std::vector<W> ws;
            for (int t = 0; t < F; ++t) {  // DB.cpp:925-927
                u64 a = (N * static_cast<u64>(t)) / static_cast<u64>(F), b = (N * static_cast<u64>(t + 1)) / static_cast<u64>(F);
                u64 step = static_cast<u64>(std::max(3, static_cast<int>((b - a) / static_cast<u64>(P.clt.base / F))));
                // group 0 is the LEADER — fast worker 0, the one whose own statistics decide rule A (DB.cpp:936-961: a fast
                // thread judges its own samples); every other worker, fast or slow, is group 1
                ws.push_back({a, step, prog_count(a, b, step), t == 0 ? 0u : 1u});
            }
            for (int t = 0; t < S; ++t) {  // DB.cpp:983-990
                u64 a = (N * static_cast<u64>(t)) / static_cast<u64>(S), b = (N * static_cast<u64>(t + 1)) / static_cast<u64>(S);
                u64 step = static_cast<u64>(std::max(1, static_cast<int>((b - a) / static_cast<u64>(P.clt.base / S))));
                u64 first = a + step / 2;
                ws.push_back({first, step, prog_count(first, b, step), 1});
            }
            for (const auto& w : ws) { P.clt.max_count = std::max(P.clt.max_count, w.count); P.global_samples += w.count; }
            u64 R = q.clt_round0 ? q.clt_round0 : static_cast<u64>(q.check_interval);
            u64 g = q.clt_growth ? q.clt_growth : 1;
            // A fast and a slow pointer that sweep the same region with the same step (the usual case:
            // equal thread counts, region/(base/F) >= 3) are emitted as ONE pair family so the GPU reads
            // each cache line of the region once for both.
            const bool pairable = (F == S);
            u64 b0 = 0;
            while (b0 < P.clt.max_count) {
                u64 b1 = (R > P.clt.max_count - b0) ? P.clt.max_count : b0 + R;
                std::vector<aqe_family> rf;
                auto window = [&](const W& w, u64& lo, u64& hi) { lo = std::min(b0, w.count); hi = std::min(b1, w.count); };
                for (int t = 0; t < T; ++t) {
                    const W& w = ws[static_cast<size_t>(t)];
                    u64 lo, hi;
                    window(w, lo, hi);
                    if (pairable && t < F && ws[static_cast<size_t>(F + t)].step == w.step) {
                        const W& v = ws[static_cast<size_t>(F + t)];
                        u64 lo2, hi2;
                        window(v, lo2, hi2);
                        if (hi <= lo && hi2 <= lo2) continue;
                        aqe_family f = strided(w.first, w.step, std::max(w.count, v.count), w.group);
                        f.ord_lo = lo; f.ord_hi = hi;
                        f.flags = AQE_F_PAIR;
                        f.row0_b = v.first; f.ord_lo_b = lo2; f.ord_hi_b = hi2;
                        clip_push(rf, f, shard);
                        continue;
                    }
                    if (pairable && t >= F && ws[static_cast<size_t>(t - F)].step == w.step) continue;  // emitted with its fast twin
                    if (hi <= lo) continue;
                    aqe_family f = strided(w.first, w.step, w.count, w.group);
                    f.ord_lo = lo;
                    f.ord_hi = hi;
                    clip_push(rf, f, shard);
                }
                P.round_fams.push_back(std::move(rf));
                b0 = b1;
                R = (R > (UINT64_MAX / 4) / g) ? UINT64_MAX / 4 : R * g;
                if (P.round_fams.size() > (1u << 20)) AQE_FAIL("clt: more than 2^20 rounds; raise clt_round0 or clt_growth");
            }
            P.rounds = static_cast<uint32_t>(P.round_fams.size());
            if (!(q.flags & AQE_Q_NO_TOPUP)) {  // DB.cpp:1031-1040
                int additional = P.clt.base / 4;
                if (additional > 0) {
                    u64 step = static_cast<u64>(std::max(1, static_cast<int>(N / static_cast<u64>(additional))));
                    aqe_family f = strided(0, step, std::min<u64>(ceil_div(N, step), static_cast<u64>(P.clt.base)));
                    f.flags = AQE_F_TOPUP;
                    P.has_topup = true;
                    // the device limits ord_hi to base - collected, which is a prefix of the ordinals, so
                    // clipping to the shard stays valid: a shard keeps ordinals [a,b) and the device
                    // intersects with [0, limit).
                    clip_push(P.topup_fams, f, shard);
                }
            }
            return AQE_OK;
        }
        default:
            AQE_FAIL("unknown method");
    }
}

bool parse_where(const char* query, double* lo, double* hi) {
    // same three shapes, same precedence and the same case sensitivity as SCH.cpp:277-294
    static const std::regex between(R"(amount\s+BETWEEN\s+(\d+(?:\.\d+)?)\s+AND\s+(\d+(?:\.\d+)?))");
    static const std::regex range(R"(amount\s*>=\s*(\d+(?:\.\d+)?)\s+AND\s+amount\s*<=\s*(\d+(?:\.\d+)?))");
    static const std::regex greater(R"(amount\s*>\s*(\d+(?:\.\d+)?))");
    std::string s(query ? query : "");
    std::smatch m;
    if (std::regex_search(s, m, between) || std::regex_search(s, m, range)) {
        *lo = std::stod(m[1]);
        *hi = std::stod(m[2]);
        return true;
    }
    if (std::regex_search(s, m, greater)) {
        *lo = std::stod(m[1]);
        *hi = 99999.99;  // SCH.cpp:290 default upper bound; note the bound stays inclusive
        return true;
    }
    *lo = -1;
    *hi = -1;
    return false;
}

double confidence_heuristic(double pct, uint64_t total) {
    double sample = static_cast<double>(total) * pct / 100.0;
    if (sample >= 1000) return 0.95;
    if (sample >= 500) return 0.90;
    if (sample >= 100) return 0.85;
    if (sample >= 50) return 0.80;
    return 0.70;
}

double error_to_sample_percent(double e) {
    if (e <= 1.0) return 20.0;
    if (e <= 2.0) return 15.0;
    if (e <= 5.0) return 10.0;
    return 5.0;
}

}  // namespace aqe
