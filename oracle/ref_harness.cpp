// oracle/ref_harness.cpp — TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// A C-ABI driver around the *reference's own* C++ hot path, compiled by oracle/Makefile from
// the sources where they lie under /root/reference (nothing is copied into this repo):
//     $(REF)/src/aqe_backend/core/custom_bplus_db.cpp      (samplers, reducers, CLT monitor)
//     $(REF)/src/aqe_backend/core/custom_scheduler.cpp     (façade, WHERE regex, confidence)
// Output goes only to oracle/_ref/ (git-ignored).  It is used for exactly two things:
//   (1) oracle/make_golden.py runs it here to produce tests/golden/*.json, which pin the
//       C restatement in oracle/aqe_oracle.c;
//   (2) bench.py's cpu_baseline leg times it on the GPU box's host cores (kind="reference").
//
// This file contains no reference code: it includes the reference headers by path and calls the
// reference's public methods.  It is built with -fno-access-control for one purpose — the
// reference's only population path, insert_record, rebuilds the whole flat cache every 1000th
// row (custom_bplus_db.cpp:188-191, O(N^2)); ref_fill_direct() instead fills one leaf in O(N).
// Every in-scope sampler sees the table only through collect_leaf_records()
// (custom_bplus_db.cpp:715-735) or cached_records_, so its outputs are unchanged; ref_fill_insert()
// keeps the slow, fully faithful path for small N so tests can check that equivalence.
#include "custom_bplus_db.hpp"
#include "custom_scheduler.hpp"

#include <cstdint>
#include <cstring>
#include <exception>
#include <random>
#include <string>
#include <utility>
#include <vector>

static_assert(sizeof(Record) == 32, "reference Record is 32 bytes (custom_bplus_db.hpp:17-27)");

namespace {

struct Handle {
    CustomBPlusDB db;
    std::vector<Record> last;  // result of the most recent sampler call
};

// cache policy of insert_record (custom_bplus_db.cpp:186-191): the flat cache is refreshed only when
// total_records % 1000 == 0, so after N inserts it holds floor(N/1000)*1000 rows.
void mimic_cache_policy(CustomBPlusDB& db, const Record* rows, size_t n) {
    size_t m = (n / 1000) * 1000;
    db.cached_records_.assign(rows, rows + m);
    db.memory_mapped_ = m > 0;
}

}  // namespace

extern "C" {

void* ref_create() { return new Handle(); }

void ref_destroy(void* h) {
    auto* H = static_cast<Handle*>(h);
    H->db.db_path_.clear();  // never let the destructor auto-save (custom_bplus_db.cpp:157-162)
    delete H;
}

// O(N) fill: one giant leaf + the cache state insert_record would have left behind.
int ref_fill_direct(void* h, const void* rows32, uint64_t n) {
    auto* H = static_cast<Handle*>(h);
    const Record* rows = static_cast<const Record*>(rows32);
    auto leaf = std::make_shared<BPlusTreeNode>(true);
    leaf->records.assign(rows, rows + n);
    leaf->keys.resize(n);
    for (uint64_t i = 0; i < n; ++i) leaf->keys[i] = rows[i].id;
    leaf->key_count = static_cast<int>(n);
    leaf->subtree_record_count = n;
    H->db.root = leaf;
    H->db.total_records = n;
    H->db.tree_height = 1;
    mimic_cache_policy(H->db, rows, n);
    return 0;
}

// Faithful fill through the public API (quadratic; small N only).
int ref_fill_insert(void* h, const void* rows32, uint64_t n) {
    auto* H = static_cast<Handle*>(h);
    const Record* rows = static_cast<const Record*>(rows32);
    for (uint64_t i = 0; i < n; ++i)
        if (!H->db.insert_record(rows[i])) return -1;
    return 0;
}

uint64_t ref_total_records(void* h) { return static_cast<Handle*>(h)->db.get_total_records(); }
uint64_t ref_cache_rows(void* h) { return static_cast<Handle*>(h)->db.cached_records_.size(); }
uint64_t ref_node_count(void* h) { return static_cast<Handle*>(h)->db.get_node_count(); }
uint64_t ref_tree_height(void* h) { return static_cast<Handle*>(h)->db.get_tree_height(); }

double ref_sum_amount(void* h) { return static_cast<Handle*>(h)->db.sum_amount(); }
double ref_avg_amount(void* h) { return static_cast<Handle*>(h)->db.avg_amount(); }
double ref_sum_amount_where(void* h, double lo, double hi) {
    return static_cast<Handle*>(h)->db.sum_amount_where(lo, hi);
}

// random_device-seeded C++ reducers: statistical parity only.
double ref_parallel_sum_sample(void* h, double pct, int t) {
    return static_cast<Handle*>(h)->db.parallel_sum_sample(pct, t);
}
double ref_parallel_avg_sample(void* h, double pct, int t) {
    return static_cast<Handle*>(h)->db.parallel_avg_sample(pct, t);
}
uint64_t ref_parallel_count_sample(void* h, double pct, int t) {
    return static_cast<Handle*>(h)->db.parallel_count_sample(pct, t);
}
double ref_parallel_sum_where_sample(void* h, double lo, double hi, double pct, int t) {
    return static_cast<Handle*>(h)->db.parallel_sum_where_sample(lo, hi, pct, t);
}
double ref_fast_aggregated_memory_stride_sum(void* h, double pct, int t) {
    return static_cast<Handle*>(h)->db.fast_aggregated_memory_stride_sum(pct, t);
}

// Run one record-returning sampler; the rows stay in the handle (ref_last_*).  Returns the number
// of rows, or -1 if the reference threw.  a,b,c,d carry the method's positional arguments.
int64_t ref_sample(void* h, int method, double pct, double a, double b, double c, double d) {
    auto* H = static_cast<Handle*>(h);
    CustomBPlusDB& db = H->db;
    try {
        switch (method) {
            case 1: H->last = db.memory_stride_sample(pct, static_cast<size_t>(a)); break;
            case 2: H->last = db.optimized_address_arithmetic_sample(pct); break;
            case 3: H->last = db.random_pointer_sample(pct, static_cast<unsigned>(a)); break;
            case 4: H->last = db.block_sample(pct, static_cast<size_t>(a)); break;
            case 5: H->last = db.page_sample(pct, static_cast<size_t>(a)); break;
            case 6: H->last = db.parallel_block_sample(pct, static_cast<size_t>(a), static_cast<int>(b)); break;
            case 7: H->last = db.optimized_clt_sample(pct, a, static_cast<int>(b), static_cast<int>(c), d); break;
            case 8: H->last = db.clt_validated_dual_pointer_sample(pct, a, static_cast<int>(b), static_cast<int>(c), d); break;
            case 9: H->last = db.fast_pointer_sample(pct, static_cast<int>(a)); break;
            case 10: H->last = db.slow_pointer_sample(pct); break;
            case 11: H->last = db.dual_pointer_sample(pct); break;
            case 12: H->last = db.parallel_pointer_sample(pct, static_cast<int>(a)); break;
            case 13: H->last = db.multithreaded_memory_stride_sample(pct, static_cast<int>(a)); break;
            case 14: H->last = db.random_start_memory_stride_sample(pct, static_cast<size_t>(a)); break;
            case 15: H->last = db.signal_based_clt_sample(pct, static_cast<int>(a)); break;
            case 16: H->last = db.sample_records(pct); break;
            case 17: H->last = db.adaptive_block_sample(pct, static_cast<size_t>(a), static_cast<size_t>(b)); break;
            case 18: H->last = db.stratified_block_sample(pct, static_cast<size_t>(a), static_cast<int>(b)); break;
            case 19: H->last = db.direct_access_sample(pct); break;         // needs the real tree: ref_fill_insert
            case 20: H->last = db.optimized_sequential_sample(pct); break;  // std::random_device start: statistical
            default: return -2;
        }
    } catch (const std::exception&) {
        H->last.clear();
        return -1;
    }
    return static_cast<int64_t>(H->last.size());
}

// Copy out the ids / amounts / whole rows of the last sample, in the order the reference returned.
int64_t ref_last_ids(void* h, int64_t* out, int64_t cap) {
    auto* H = static_cast<Handle*>(h);
    int64_t n = static_cast<int64_t>(H->last.size());
    for (int64_t i = 0; i < n && i < cap; ++i) out[i] = H->last[i].id;
    return n;
}
int64_t ref_last_amounts(void* h, double* out, int64_t cap) {
    auto* H = static_cast<Handle*>(h);
    int64_t n = static_cast<int64_t>(H->last.size());
    for (int64_t i = 0; i < n && i < cap; ++i) out[i] = H->last[i].amount;
    return n;
}
int64_t ref_last_rows(void* h, void* out32, int64_t cap) {
    auto* H = static_cast<Handle*>(h);
    int64_t n = static_cast<int64_t>(H->last.size());
    int64_t m = n < cap ? n : cap;
    if (m > 0) std::memcpy(out32, H->last.data(), static_cast<size_t>(m) * sizeof(Record));
    return n;
}

// key_count of every leaf in leaf order (leftmost leaf, then the next_leaf chain): pins the restatement's leaf model
int64_t ref_leaf_sizes(void* h, uint32_t* out, int64_t cap) {
    auto* H = static_cast<Handle*>(h);
    auto node = H->db.root;
    while (node && !node->is_leaf) node = node->children.empty() ? nullptr : node->children[0];
    int64_t n = 0;
    for (; node; node = node->next_leaf, ++n)
        if (out && n < cap) out[n] = static_cast<uint32_t>(node->key_count);
    return n;
}

// this container's libstdc++: what optimized_sequential_sample's start offset would be were its generator mt19937(seed)
// (the reference seeds it from std::random_device).  No reference code involved: <random> only.
double ref_uniform_real(uint32_t seed, double hi) {
    std::mt19937 gen(seed);
    std::uniform_real_distribution<> dis(0.0, hi);
    return dis(gen);
}

int ref_save_to_file(void* h, const char* path) {
    return static_cast<Handle*>(h)->db.save_to_file(path) ? 1 : 0;
}

// ---- façade (custom_scheduler.cpp): deterministic helpers reached with -fno-access-control ----
double ref_sched_confidence(double pct, uint64_t total) {
    CustomApproximateScheduler s(0.05);
    double v = s.calculate_confidence_level(pct, total);
    s.db_->db_path_.clear();
    return v;
}
// returns 1 and fills lo/hi; {-1,-1} means "no WHERE" (custom_scheduler.cpp:277-294)
int ref_sched_where(const char* query, double* lo, double* hi) {
    CustomApproximateScheduler s(0.05);
    auto p = s.extract_where_conditions(query);
    *lo = p.first;
    *hi = p.second;
    s.db_->db_path_.clear();
    return 1;
}

}  // extern "C"
