// tests/c_host/planner_fuzz.cpp — the host planner (csrc/planner.cpp) under AddressSanitizer + UBSan (CPU only; GPU sanitizers
// are not available on the pool).  Seeded random queries of every sampler, hostile parameters included; for every plan that
// is accepted:
//   * every family row lies inside the table and inside the shard it was clipped to,
//   * the shards of a partition take, together, exactly the rows of the whole-table plan (row checksum + count),
//   * global_samples equals the whole-table plan's rows.
// Built and run by tests/test_host_planner.py::test_planner_under_sanitizers.
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../approximatequeryengine_amd/csrc/planner.hpp"

using namespace aqe;

static uint64_t rng_state = 1;
static uint64_t rnd() {
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t below(uint64_t n) { return n ? rnd() % n : 0; }
static double unit() { return static_cast<double>(rnd() >> 11) / 9007199254740992.0; }

struct Tally {
    uint64_t rows = 0, checksum = 0;
};

static bool walk(const std::vector<aqe_family>& fams, uint64_t N, ClipWindow w, Tally& t, uint64_t budget, std::string& why) {
    for (const aqe_family& f : fams) {
        if (f.seg_len == 0) { why = "seg_len 0"; return false; }
        auto one = [&](uint64_t row0, uint64_t lo, uint64_t hi, uint64_t seg_len, uint64_t pitch) {
            if (hi - lo > budget) { why = "skip"; return false; }
            for (uint64_t o = lo; o < hi; ++o) {
                const uint64_t r = row0 + (o / seg_len) * pitch + (o % seg_len) * f.step;
                if (r >= N || r < w.lo || r >= w.hi) {
                    char b[160];
                    snprintf(b, sizeof b, "row %" PRIu64 " outside table %" PRIu64 " / shard [%" PRIu64 ", %" PRIu64 ")", r, N, w.lo, w.hi);
                    why = b;
                    return false;
                }
                t.rows += 1;
                t.checksum += (r + 1) * 0x9E3779B97F4A7C15ull;
            }
            return true;
        };
        if (f.ord_hi > f.ord_lo && !one(f.row0, f.ord_lo, f.ord_hi, f.seg_len, f.pitch)) return false;
        if ((f.flags & AQE_F_PAIR) && f.ord_hi_b > f.ord_lo_b && !one(f.row0_b, f.ord_lo_b, f.ord_hi_b, ~0ull, 0)) return false;
    }
    return true;
}

int main(int argc, char** argv) {
    const uint64_t seed = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1;
    const int iters = argc > 2 ? atoi(argv[2]) : 2000;
    rng_state = seed * 0x2545F4914F6CDD1Dull + 7;
    int accepted = 0, refused = 0, skipped = 0;
    static const int methods[] = {0, 1, 2, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17};  // (3, 18, 19, 20: row lists, not families)
    for (int it = 0; it < iters; ++it) {
        aqe_query q;
        std::memset(&q, 0, sizeof q);
        q.method = methods[below(sizeof methods / sizeof methods[0])];
        q.num_threads = static_cast<int32_t>(below(40)) - (below(10) == 0 ? 3 : 0);
        const int pk = static_cast<int>(below(8));
        q.sample_percent = pk == 0 ? 0.0 : pk == 1 ? 100.0 : pk == 2 ? 150.0 : pk == 3 ? -5.0 : pk == 4 ? 1e-7 : 100.0 * unit();
        q.stride_bytes = below(4) ? 0 : below(100000);
        q.block_size = below(6) ? 1 + below(5000) : below(3);
        q.block_size_max = q.block_size + below(3000) - (below(9) == 0 ? 1 : 0);
        q.seed = rnd();
        q.step_size = static_cast<int32_t>(below(12)) - 2;
        q.check_interval = static_cast<int32_t>(below(40)) - 1;
        q.confidence_level = 0.8 + 0.2 * unit();
        q.max_error_percent = 5.0 * unit();
        q.clt_round0 = below(3) ? 0 : 1 + below(5000);
        q.clt_growth = static_cast<uint32_t>(below(6));
        const int nk = static_cast<int>(below(10));
        const uint64_t N = nk == 0 ? below(12) : nk == 1 ? 1000000000ull + below(1000) : nk == 2 ? (1ull << 32) + below(1000) : 1 + below(300000);
        if (below(5) == 0) q.visible_rows = below(N + 10);
        if (below(6) == 0) { q.row_lo = below(N + 2); q.row_hi = q.row_lo + below(N + 2); }
        double zv[10];
        for (double& v : zv) v = below(7) ? 1000.0 * unit() : 0.0;

        HostPlan whole;
        std::string err;
        int rc = build_plan(q, N, ClipWindow{0, N}, whole, err, zv);
        if (rc != AQE_OK) { ++refused; continue; }
        if (whole.is_random || whole.is_perm) continue;
        const uint64_t budget = 3000000;
        Tally all;
        std::string why;
        bool ok = true;
        std::vector<std::vector<aqe_family>> rounds = whole.round_fams;
        if (whole.has_topup) rounds.push_back(whole.topup_fams);
        for (const auto& rf : rounds)
            if (!walk(rf, whole.on_sorted ? N : N, ClipWindow{0, N}, all, budget, why)) { ok = false; break; }
        if (!ok && why == "skip") { ++skipped; continue; }
        if (!ok) { fprintf(stderr, "seed %" PRIu64 " it %d method %d N %" PRIu64 ": %s\n", seed, it, q.method, N, why.c_str()); return 1; }
        if (!whole.has_topup && !whole.is_clt && all.rows != whole.global_samples) {
            fprintf(stderr, "pct %.17g T %d B %" PRIu64 " Bmax %" PRIu64 " stride %" PRIu64 " visible %" PRIu64 " window [%" PRIu64 ", %" PRIu64 ") step_size %d\n", q.sample_percent,
                    q.num_threads, q.block_size, q.block_size_max, q.stride_bytes, q.visible_rows, q.row_lo, q.row_hi, q.step_size);
            fprintf(stderr, "seed %" PRIu64 " it %d method %d N %" PRIu64 ": global_samples %" PRIu64 " but %" PRIu64 " rows\n", seed, it, q.method, N,
                    whole.global_samples, all.rows);
            return 1;
        }
        // a partition into G shards (ragged, some empty) takes the same rows
        const int G = 1 + static_cast<int>(below(5));
        std::vector<uint64_t> cuts{0};
        for (int g = 1; g < G; ++g) cuts.push_back(below(N + 1));
        cuts.push_back(N);
        for (size_t i = 1; i < cuts.size(); ++i)
            for (size_t j = i; j > 0 && cuts[j] < cuts[j - 1]; --j) std::swap(cuts[j], cuts[j - 1]);
        Tally parts;
        for (int g = 0; g < G && ok; ++g) {
            HostPlan P;
            rc = build_plan(q, N, ClipWindow{cuts[g], cuts[g + 1]}, P, err, zv);
            if (rc != AQE_OK) { fprintf(stderr, "seed %" PRIu64 " it %d: shard refused what the whole table accepted: %s\n", seed, it, err.c_str()); return 1; }
            if (P.rounds != whole.rounds || P.global_samples != whole.global_samples || P.has_topup != whole.has_topup) {
                fprintf(stderr, "seed %" PRIu64 " it %d method %d: shard plan disagrees on rounds/samples\n", seed, it, q.method);
                return 1;
            }
            std::vector<std::vector<aqe_family>> rr = P.round_fams;
            if (P.has_topup) rr.push_back(P.topup_fams);
            for (const auto& rf : rr)
                if (!walk(rf, N, ClipWindow{cuts[g], cuts[g + 1]}, parts, budget, why)) { ok = false; break; }
        }
        if (!ok) { fprintf(stderr, "seed %" PRIu64 " it %d method %d N %" PRIu64 " (sharded): %s\n", seed, it, q.method, N, why.c_str()); return 1; }
        if (parts.rows != all.rows || parts.checksum != all.checksum) {
            fprintf(stderr, "seed %" PRIu64 " it %d method %d N %" PRIu64 ": %d shards take %" PRIu64 " rows, the whole table %" PRIu64 "\n", seed, it, q.method, N, G,
                    parts.rows, all.rows);
            return 1;
        }
        ++accepted;
    }
    // the other host entry points on hostile input
    double lo, hi;
    const char* texts[] = {"", "WHERE", "select sum(amount) from sales where amount between 1 and", "WHERE amount BETWEEN 1e400 AND -1e400",
                           "where amount between 250 and 750", "WHERE amount BETWEEN AND", "WHERE amount >= 5 AND amount <= 1"};
    for (const char* t : texts) (void)parse_where(t, &lo, &hi);
    for (double p : {0.0, -1.0, 1e300, 50.0}) { (void)confidence_heuristic(p, 0); (void)confidence_heuristic(p, ~0ull); (void)error_to_sample_percent(p); }
    std::vector<uint64_t> idx;
    std::string err;
    for (uint64_t N : {0ull, 1ull, 1000ull, 100000ull})
        for (double p : {0.0, 0.5, 100.0, 250.0, -3.0}) (void)random_pointer_indices(N, p, static_cast<uint32_t>(rnd()), ClipWindow{N / 3, N}, idx, err);
    printf("planner_fuzz ok: seed %" PRIu64 ", %d accepted, %d refused, %d too large to walk\n", seed, accepted, refused, skipped);
    return 0;
}
