// planner.hpp — host-side query planning: turns (method, parameters, table size, shard) into the short
// list of arithmetic row families the GPU kernels sweep.  Pure C++, no HIP: exported through the C ABI
// (aqe_plan_families / aqe_plan_random_indices) so it is testable without a GPU.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/aqe_hip.h"

namespace aqe {

struct ClipWindow {
    uint64_t lo, hi;  // shard rows [lo, hi) in global row numbering
};

struct CltShape {
    int base = 0;        // int(N*pct/100), DB.cpp:896
    int n_workers = 0;   // T
    int n_fast = 0;      // T/2, DB.cpp:918
    double z = 1.96;     // DB.cpp:911-912
    double e = 0.0;      // max_error_percent
    uint64_t max_count = 0;  // longest worker progression
};

struct HostPlan {
    bool is_random = false;   // RANDOM_POINTER: explicit index list instead of families
    bool is_perm = false;     // RANDOM_DEVICE: rows = perm_lo + P(k), k < perm_target (PermSpec below), no list at all
    uint64_t perm_n = 0, perm_lo = 0, perm_target = 0, perm_seed = 0;
    bool is_clt = false;
    bool has_topup = false;
    bool on_sorted = false;   // STRATIFIED_BLOCK: families index the amount-sorted table
    uint32_t rounds = 1;
    CltShape clt;
    std::vector<std::vector<aqe_family>> round_fams;  // [round] -> families clipped to the shard
    std::vector<aqe_family> topup_fams;               // CLT top-up (flag AQE_F_TOPUP)
    std::vector<uint64_t> random_idx;                 // ascending global rows inside the shard
    uint64_t global_samples = 0;  // samples the whole query draws over the whole table (all rounds)
    uint64_t visible_rows = 0;    // M actually used
    double pct = 0.0;
};

// The keyed bijection of AQE_M_RANDOM_DEVICE on [0, n): x -> ((x + k0) C1 ^>> s1 + k1) C2 ^>> s2, C3 ^>> s3, C1 ^>> s1 on
// `bits` = ceil(log2 n) bits (every step is a bijection of the bits-bit integers: add, multiply by an odd constant,
// xor with a right shift), repeated while the result is >= n (cycle walking).  Host and device share this definition.
struct PermSpec {
    uint64_t n, lo, target, k0, k1, mask;
    uint32_t s1, s2, s3, pad;
};
constexpr uint64_t kPermC1 = 0x9E3779B97F4A7C15ull, kPermC2 = 0xBF58476D1CE4E5B9ull, kPermC3 = 0x94D049BB133111EBull;
inline uint64_t perm_splitmix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
inline PermSpec perm_spec(uint64_t n, uint64_t lo, uint64_t target, uint64_t seed) {
    PermSpec p{};
    uint32_t bits = 1;
    while (bits < 64 && (n - 1) >> bits) ++bits;  // ceil(log2 n), at least 1
    p.n = n; p.lo = lo; p.target = target;
    p.k0 = perm_splitmix64(seed);
    p.k1 = perm_splitmix64(seed ^ 0xA5A5A5A5A5A5A5A5ull);
    p.mask = bits >= 64 ? ~0ull : ((1ull << bits) - 1);
    p.s1 = bits / 2 ? bits / 2 : 1;
    p.s2 = bits / 3 ? bits / 3 : 1;
    p.s3 = (2 * bits) / 3 ? (2 * bits) / 3 : 1;
    return p;
}

// Appends the part of family f whose rows lie in [w.lo, w.hi) (nothing when it has none).
void clip_family(std::vector<aqe_family>& out, const aqe_family& f, ClipWindow w);

// Returns AQE_OK or AQE_ERR_INVALID (err gets the reason).  shard = [lo, hi) global rows.
// zone_var: the ten zone variances ADAPTIVE_BLOCK needs (nullptr for every other method).
int build_plan(const aqe_query& q, uint64_t n_global, ClipWindow shard, HostPlan& out, std::string& err,
               const double* zone_var = nullptr);

// random_pointer_sample index set (DB.cpp:856-882), ascending, restricted to the shard.
int random_pointer_indices(uint64_t n_global, double pct, uint32_t seed, ClipWindow shard,
                           std::vector<uint64_t>& out, std::string& err);

// total ordinals in a family window
inline uint64_t family_size(const aqe_family& f) {
    uint64_t n = f.ord_hi > f.ord_lo ? f.ord_hi - f.ord_lo : 0;
    if ((f.flags & AQE_F_PAIR) && f.ord_hi_b > f.ord_lo_b) n += f.ord_hi_b - f.ord_lo_b;
    return n;
}

// row of ordinal o
inline uint64_t family_row(const aqe_family& f, uint64_t o) {
    return f.row0 + (o / f.seg_len) * f.pitch + (o % f.seg_len) * f.step;
}

bool parse_where(const char* query, double* lo, double* hi);   // SCH.cpp:277-294
double confidence_heuristic(double pct, uint64_t total);        // SCH.cpp:296-305
double error_to_sample_percent(double e);                       // CLI:243-250

}  // namespace aqe
