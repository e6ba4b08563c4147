/* oracle/aqe_oracle.c — TEST INFRASTRUCTURE ONLY (see aqe_oracle.h for the rules of use).
 *
 * Plain-C restatement of the reference's sampled-reduce path.  Nothing here is copied: each routine
 * re-derives the index arithmetic from the cited reference lines and is pinned against the compiled
 * reference through tests/golden/.  Integer expressions deliberately keep the reference's types
 * (int target counts, size_t strides, left-to-right double products) because truncation is part of
 * the behaviour being pinned.
 */
#include "aqe_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Synthetic table (value distribution of src/aqe_frontend/utils.py:43, uniform(1,1000))
 * ---------------------------------------------------------------------------------------------- */
uint64_t aqo_splitmix64_at(uint64_t seed, uint64_t i) {
    uint64_t z = seed + (i + 1) * 0x9E3779B97F4A7C15ULL; /* state after i+1 increments */
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

double aqo_synth_amount(uint64_t seed, uint64_t i) {
    double u = (double)(aqo_splitmix64_at(seed, i) >> 11) * (1.0 / 9007199254740992.0);
    return 1.0 + 999.0 * u;
}

void aqo_synth_fill(aqo_record* rows, uint64_t first_row, uint64_t n, uint64_t seed) {
    for (uint64_t k = 0; k < n; ++k) {
        uint64_t i = first_row + k;
        rows[k].id = (int64_t)(i + 1);
        rows[k].amount = aqo_synth_amount(seed, i);
        rows[k].region = (int32_t)(i % 4);
        rows[k].product_id = (int32_t)(i % 100);
        rows[k].timestamp = (int64_t)i;
    }
}

/* ------------------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint64_t* out;
    int64_t cap;
    int64_t n;
} sink;

static inline void emit(sink* s, uint64_t idx) {
    if (s->out && s->n < s->cap) s->out[s->n] = idx;
    s->n++;
}

/* static_cast<int>(size * sample_percent / 100.0): the product is formed first (DB.cpp:1542 etc.) */
static inline int target_of(uint64_t rows, double pct) { return (int)((double)rows * pct / 100.0); }

static inline uint64_t umax(uint64_t a, uint64_t b) { return a > b ? a : b; }
static inline uint64_t umin(uint64_t a, uint64_t b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------------------------------
 * R2 — strided samplers
 * ---------------------------------------------------------------------------------------------- */
/* DB.cpp:1540-1566 (cached path) == DB.cpp:1569-1602 (fallback) for index arithmetic.  M is the number
 * of rows the sampler sees: the flat cache length (floor(N/1000)*1000 after plain inserts, DB.cpp:188). */
int64_t aqo_idx_memory_stride(uint64_t M, double pct, uint64_t stride_bytes, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (M == 0) return 0;
    int target = target_of(M, pct);
    if (target <= 0) return 0;
    uint64_t stride = stride_bytes == 0 ? umax(1, M / (uint64_t)target)
                                        : umax(1, stride_bytes / sizeof(aqo_record));
    for (uint64_t off = 0; (uint64_t)s.n < (uint64_t)target && off < M; off += stride) emit(&s, off);
    return s.n;
}

/* DB.cpp:1838-1878: memory_stride_sample from a random first row in [0, stride).  The reference draws the
 * start from mt19937(random_device); start_override replays a recorded draw, otherwise the start is the
 * product's seeded one, splitmix64_at(seed, 0) % stride. */
int64_t aqo_idx_random_start_stride(uint64_t M, double pct, uint64_t stride_bytes, uint64_t seed,
                                    const uint64_t* start_override, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (M == 0) return 0;
    int target = target_of(M, pct);
    if (target <= 0) return 0;
    uint64_t stride = stride_bytes == 0 ? umax(1, M / (uint64_t)target)
                                        : umax(1, stride_bytes / sizeof(aqo_record));
    uint64_t start = start_override ? *start_override : aqo_splitmix64_at(seed, 0) % stride;
    for (uint64_t off = start; (uint64_t)s.n < (uint64_t)target && off < M; off += stride) emit(&s, off);
    return s.n;
}

/* DB.cpp:1667-1703: indices i*stride, i < target, kept when < M. */
int64_t aqo_idx_address_arithmetic(uint64_t M, double pct, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (M == 0) return 0;
    int target = target_of(M, pct);
    if (target <= 0) return 0;
    uint64_t stride = M / (uint64_t)target;
    if (stride == 0) stride = 1;
    for (int i = 0; i < target; ++i) {
        uint64_t off = (uint64_t)i * stride;
        if (off < M) emit(&s, off);
    }
    return s.n;
}

/* DB.cpp:737-758 (fast_pointer_sample); step_size=1 gives slow_pointer_sample (DB.cpp:760-780). */
int64_t aqo_idx_fast_pointer(uint64_t N, double pct, int step_size, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    int target = target_of(N, pct);
    if (target <= 0) return 0;
    int step = imax(1, (int)(N / (uint64_t)target));
    step *= step_size;
    if (step <= 0) return -1; /* the reference would loop forever / walk backwards */
    for (uint64_t i = 0; i < N && s.n < target; i += (uint64_t)step) emit(&s, i);
    return s.n;
}

/* DB.cpp:782-813. */
int64_t aqo_idx_dual_pointer(uint64_t N, double pct, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    int target = target_of(N, pct);
    if (target <= 0) return 0;
    int fast_target = target / 3;
    int slow_target = target - fast_target;
    if (fast_target == 0) return -1; /* N / fast_target: integer division by zero */
    int fast_step = imax(1, (int)(N / (uint64_t)fast_target)) * 3;
    for (uint64_t i = 0; i < N && s.n < fast_target; i += (uint64_t)fast_step) emit(&s, i);
    int slow_step = imax(1, (int)(N / (uint64_t)slow_target));
    uint64_t offset = (uint64_t)(fast_step / 2);
    for (uint64_t i = offset; i < N && s.n < target; i += (uint64_t)slow_step) emit(&s, i);
    return s.n;
}

/* DB.cpp:815-854: thread t starts at (N/T)*t, global step, per-thread cap target/T; thread order. */
int64_t aqo_idx_parallel_pointer(uint64_t N, double pct, int T, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    int target = target_of(N, pct);
    if (target <= 0) return 0;
    if (T <= 0) return -1;
    int per_thread = target / T;
    int step = imax(1, (int)(N / (uint64_t)target));
    for (int t = 0; t < T; ++t) {
        uint64_t start = (N / (uint64_t)T) * (uint64_t)t;
        int got = 0;
        for (uint64_t i = start; i < N && got < per_thread; i += (uint64_t)step) {
            emit(&s, i);
            ++got;
        }
    }
    return s.n;
}

/* ------------------------------------------------------------------------------------------------
 * R3 — seeded random sampler: std::mt19937 + libstdc++-11 uniform_int_distribution<size_t>
 * (bits/uniform_int_dist.h:246-317: a 32-bit generator and a range < 2^32 take Lemire's
 * nearly-divisionless method on 64-bit products) + std::set until `target` unique (DB.cpp:856-882).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t mt[624];
    int idx;
} mt19937;

static void mt_seed(mt19937* g, uint32_t seed) {
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

static uint32_t mt_next(mt19937* g) {
    if (g->idx >= 624) {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (g->mt[k] & 0x80000000u) | (g->mt[(k + 1) % 624] & 0x7fffffffu);
            g->mt[k] = g->mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

void aqo_mt19937_stream(uint32_t seed, uint32_t* out, int n) {
    mt19937 g;
    mt_seed(&g, seed);
    for (int i = 0; i < n; ++i) out[i] = mt_next(&g);
}

static uint32_t lemire32(mt19937* g, uint32_t range) {
    uint64_t product = (uint64_t)mt_next(g) * (uint64_t)range;
    uint32_t low = (uint32_t)product;
    if (low < range) {
        uint32_t threshold = (uint32_t)(0u - range) % range;
        while (low < threshold) {
            product = (uint64_t)mt_next(g) * (uint64_t)range;
            low = (uint32_t)product;
        }
    }
    return (uint32_t)(product >> 32);
}

int64_t aqo_idx_random_pointer(uint64_t N, double pct, uint32_t seed, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    int target = target_of(N, pct);
    if (target <= 0) return 0;
    if (N > 0xFFFFFFFFull) return -1; /* 64-bit ranges take a different libstdc++ branch; out of scope */
    uint64_t want = umin((uint64_t)target, N);
    uint8_t* bits = (uint8_t*)calloc((size_t)((N + 7) / 8), 1);
    if (!bits) return -1;
    mt19937 g;
    mt_seed(&g, seed);
    uint64_t have = 0;
    while (have < want) {
        /* N == 2^32-1+1 cannot happen (N <= 2^32-1); range N <= 2^32-1 keeps the downscaling branch,
         * except N-1 == 2^32-1 which we excluded above. */
        uint64_t v = lemire32(&g, (uint32_t)N);
        if (!(bits[v >> 3] & (1u << (v & 7)))) {
            bits[v >> 3] |= (uint8_t)(1u << (v & 7));
            ++have;
        }
    }
    for (uint64_t i = 0; i < N; ++i) /* std::set iterates ascending */
        if (bits[i >> 3] & (1u << (i & 7))) emit(&s, i);
    free(bits);
    return s.n;
}

/* ------------------------------------------------------------------------------------------------
 * R2's small-table siblings — what the reference CLI takes below 50 k rows (enhanced_aqe_cli.py:181-186)
 * ---------------------------------------------------------------------------------------------- */
/* Leaf sizes of the reference's B+ tree after N ascending inserts, by SIMULATING the inserts (DB.cpp:164-240 with
 * BPlusTreeNode::split, DB.cpp:43-62): the rightmost leaf takes every row; at MAX_KEYS = 255 keys it splits, keeps
 * MAX_KEYS / 2 = 127 and hands the rest to a new right sibling.  Returns the number of leaves (sizes[] may be NULL). */
int64_t aqo_leaf_sizes(uint64_t N, uint32_t* sizes, int64_t cap) {
    int64_t leaves = N ? 1 : 0;
    uint32_t cur = 0;
    for (uint64_t i = 0; i < N; ++i) {
        ++cur;
        if (cur >= 255) { /* insert_into_node reports a full leaf; the parent splits it */
            if (sizes && leaves - 1 < cap) sizes[leaves - 1] = 255 / 2;
            cur -= 255 / 2;
            ++leaves;
        }
    }
    if (N && sizes && leaves - 1 < cap) sizes[leaves - 1] = cur;
    return leaves;
}

/* direct_access_sample, DB.cpp:584-644: rows in the reference's order (duplicates included). */
int64_t aqo_idx_direct_access(uint64_t N, double pct, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0 || !(pct > 0.0)) return 0;
    if (pct >= 100.0) { for (uint64_t i = 0; i < N; ++i) emit(&s, i); return s.n; }
    uint64_t target = (uint64_t)((double)N * pct / 100.0); /* DB.cpp:593 */
    int64_t L = aqo_leaf_sizes(N, NULL, 0);
    uint32_t* sz = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)L);
    uint64_t* first = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)L);
    if (!sz || !first) { free(sz); free(first); return -1; }
    aqo_leaf_sizes(N, sz, L);
    uint64_t acc = 0;
    for (int64_t l = 0; l < L; ++l) { first[l] = acc; acc += sz[l]; }
    uint64_t nodes_to_sample = umax(1, target / 10);                 /* DB.cpp:621 */
    double node_step = (double)L / (double)nodes_to_sample;           /* DB.cpp:622 */
    uint64_t count = 0;
    for (uint64_t i = 0; i < nodes_to_sample && count < target; ++i) {
        uint64_t node_index = (uint64_t)((double)i * node_step);
        if (node_index < (uint64_t)L) {
            int key_count = (int)sz[node_index];
            int records_per_node = (int)(target / nodes_to_sample);
            if (records_per_node < 1) records_per_node = 1;
            if (records_per_node > key_count) records_per_node = key_count;
            double record_step = (double)key_count / records_per_node;
            for (int j = 0; j < records_per_node && count < target; ++j) {
                int record_index = (int)(j * record_step);
                if (record_index < key_count) { emit(&s, first[node_index] + (uint64_t)record_index); ++count; }
            }
        }
    }
    free(sz);
    free(first);
    return s.n;
}

/* libstdc++ 11: std::uniform_real_distribution<double>(0, hi)(std::mt19937(seed)) = generate_canonical<double, 53>
 * (bits/random.tcc: k = 2 draws, sum = g0 + g1 * 2^32, / 2^64, clamped below 1) * (hi - 0) + 0. */
double aqo_uniform_real(uint32_t seed, double hi) {
    mt19937 g;
    mt_seed(&g, seed);
    double sum = (double)mt_next(&g);
    sum += (double)mt_next(&g) * 4294967296.0;
    double r = sum / 18446744073709551616.0;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r * hi;
}

/* optimized_sequential_sample, DB.cpp:366-428, walked record by record as the reference walks its leaves; the start offset
 * (std::random_device there) from mt19937(seed). */
int64_t aqo_idx_optimized_sequential(uint64_t N, double pct, uint32_t seed, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0 || !(pct > 0.0)) return 0;
    if (pct >= 100.0) { for (uint64_t i = 0; i < N; ++i) emit(&s, i); return s.n; }
    uint64_t target = (uint64_t)((double)N * pct / 100.0);
    if (target == 0) return 0;
    double step = 100.0 / pct;
    double next_sample_point = aqo_uniform_real(seed, step);
    uint64_t current_count = 0, taken = 0;
    for (uint64_t i = 0; i < N && taken < target; ++i) {
        current_count++;
        if ((double)current_count >= next_sample_point && taken < target) {
            emit(&s, i);
            ++taken;
            next_sample_point += step;
        }
    }
    return s.n;
}

/* ------------------------------------------------------------------------------------------------
 * R4 — block family
 * ---------------------------------------------------------------------------------------------- */
/* DB.cpp:1151-1181. */
int64_t aqo_idx_block(uint64_t N, double pct, uint64_t B, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    int target = target_of(N, pct);
    if (target <= 0) return 0;
    if (B == 0) return -1;
    uint64_t total_blocks = (N + B - 1) / B;
    uint64_t to_sample = umax(1, (uint64_t)((double)total_blocks * pct / 100.0));
    uint64_t interval = total_blocks / to_sample;
    if (interval == 0) interval = 1;
    for (uint64_t blk = 0; blk < total_blocks && s.n < target; blk += interval) {
        uint64_t a = blk * B, b = umin(a + B, N);
        for (uint64_t i = a; i < b && s.n < target; ++i) emit(&s, i);
    }
    return s.n;
}

/* DB.cpp:1183-1216: a "page" is page_bytes / sizeof(Record) rows. */
int64_t aqo_idx_page(uint64_t N, double pct, uint64_t page_bytes, uint64_t* out, int64_t cap) {
    uint64_t rows_per_page = page_bytes / sizeof(aqo_record);
    if (rows_per_page == 0) rows_per_page = 1;
    return aqo_idx_block(N, pct, rows_per_page, out, cap);
}

/* DB.cpp:1218-1271: logical blocks split between threads, per-thread cap target/T, thread order. */
int64_t aqo_idx_parallel_block(uint64_t N, double pct, uint64_t B, int T, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    int target = target_of(N, pct);
    if (target <= 0) return 0;
    if (B == 0 || T <= 0) return -1;
    uint64_t total_blocks = (N + B - 1) / B;
    uint64_t to_sample = umax(1, (uint64_t)((double)total_blocks * pct / 100.0));
    uint64_t per_thread = to_sample / (uint64_t)T;
    if (per_thread == 0) per_thread = 1;
    uint64_t interval = total_blocks / to_sample;
    if (interval == 0) interval = 1;
    uint64_t thread_target = (uint64_t)(target / T);
    for (int t = 0; t < T; ++t) {
        uint64_t got = 0;
        uint64_t b0 = (uint64_t)t * per_thread, b1 = umin(b0 + per_thread, to_sample);
        for (uint64_t lb = b0; lb < b1 && got < thread_target; ++lb) {
            uint64_t a = lb * interval * B, b = umin(a + B, N);
            for (uint64_t i = a; i < b && got < thread_target; ++i) {
                emit(&s, i);
                ++got;
            }
        }
    }
    return s.n;
}

/* DB.cpp:1273-1329 — adaptive_block_sample: ten zones, per-zone population variance from raw moments, block
 * size shrinking with variance, the first max(1, len*pct/100) rows of every block.  zone_var_out (optional, 10
 * doubles) receives the zone variances.  -1 where the reference's arithmetic is undefined. */
int64_t aqo_idx_adaptive_block(const aqo_record* rows, uint64_t N, double pct, uint64_t min_block, uint64_t max_block,
                               double* zone_var_out, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    int target = target_of(N, pct);
    if (target <= 0) return 0;
    const uint64_t zones = 10, zone_size = N / zones;
    if (zone_size == 0 || max_block < min_block || min_block == 0) return -1;
    double var[10], vmax = 0.0;
    for (uint64_t z = 0; z < zones; ++z) {
        uint64_t a = z * zone_size, b = umin(a + zone_size, N);
        double sum = 0.0, sum_sq = 0.0;
        for (uint64_t i = a; i < b; ++i) { sum += rows[i].amount; sum_sq += rows[i].amount * rows[i].amount; }
        double cnt = (double)(b - a), mean = sum / cnt;
        var[z] = (sum_sq / cnt) - (mean * mean);
        if (z == 0 || var[z] > vmax) vmax = var[z];
        if (zone_var_out) zone_var_out[z] = var[z];
    }
    if (!(vmax > 0.0)) return -1; /* 0/0 in the reference */
    for (uint64_t z = 0; z < zones && s.n < target; ++z) {
        uint64_t a = z * zone_size, b = umin(a + zone_size, N);
        double ratio = var[z] / vmax;
        uint64_t abs_ = min_block + (uint64_t)((double)(max_block - min_block) * (1.0 - ratio));
        for (uint64_t i = a; i < b && s.n < target; i += abs_) {
            uint64_t be = umin(i + abs_, b);
            uint64_t cnt = umax(1, (uint64_t)((double)(be - i) * pct / 100.0));
            for (uint64_t j = 0; j < cnt && i + j < be && s.n < target; ++j) emit(&s, i + j);
        }
    }
    return s.n;
}

typedef struct { double amount; uint64_t row; } amt_row;
static int cmp_amt_row(const void* a, const void* b) {
    const amt_row* x = (const amt_row*)a; const amt_row* y = (const amt_row*)b;
    if (x->amount < y->amount) return -1;
    if (x->amount > y->amount) return 1;
    return x->row < y->row ? -1 : x->row > y->row;
}

/* DB.cpp:1331-1379 — stratified_block_sample: rows sorted by amount, `strata` equal strata, evenly spaced
 * blocks inside each, min(target/strata, block) rows from the head of every sampled block.  Emits ROW indices
 * (ties in amount are ordered by row; the reference's std::sort leaves tie order unspecified). */
int64_t aqo_idx_stratified_block(const aqo_record* rows, uint64_t N, double pct, uint64_t B, int strata,
                                 uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    int target = target_of(N, pct);
    if (target <= 0) return 0;
    if (B == 0 || strata <= 0) return -1;
    amt_row* v = (amt_row*)malloc(sizeof(amt_row) * (size_t)N);
    if (!v) return -1;
    for (uint64_t i = 0; i < N; ++i) { v[i].amount = rows[i].amount; v[i].row = i; }
    qsort(v, (size_t)N, sizeof(amt_row), cmp_amt_row);
    uint64_t stratum_size = N / (uint64_t)strata, per_stratum = (uint64_t)(target / strata);
    for (int st = 0; st < strata && s.n < target; ++st) {
        uint64_t a = (uint64_t)st * stratum_size, b = (st == strata - 1) ? N : a + stratum_size;
        uint64_t recs = b - a, blocks = (recs + B - 1) / B;
        uint64_t to_sample = umax(1, (uint64_t)((double)blocks * pct / 100.0));
        uint64_t interval = blocks / to_sample;
        if (interval == 0) interval = 1;
        for (uint64_t blk = 0; blk < blocks && s.n < target; blk += interval) {
            uint64_t bs = a + blk * B, be = umin(bs + B, b);
            uint64_t remaining = umin(per_stratum, (uint64_t)target - (uint64_t)s.n);
            uint64_t take = umin(remaining, be - bs);
            for (uint64_t i = 0; i < take; ++i) emit(&s, v[bs + i].row);
        }
    }
    free(v);
    return s.n;
}

/* ------------------------------------------------------------------------------------------------
 * R9 — optimized_clt_sample (DB.cpp:1046-1147): region-per-thread strided; the "CLT check" after the
 * loop returns the same vector on both branches, so the sampler is deterministic.
 * ---------------------------------------------------------------------------------------------- */
int64_t aqo_idx_optimized_clt(uint64_t N, double pct, int T, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (N == 0) return 0;
    uint64_t target = (uint64_t)((double)N * pct / 100.0);
    if (target == 0) return 0;
    int opt = T < imax(1, (int)(target / 100)) ? T : imax(1, (int)(target / 100));
    if (N < 5000 || target < 200 || opt == 1) {
        uint64_t step = umax(1, N / target);
        for (uint64_t i = 0; i < N && (uint64_t)s.n < target; i += step) emit(&s, i);
        return s.n;
    }
    if (opt <= 0) return 0; /* T <= 0: no workers are launched */
    uint64_t per_thread = target / (uint64_t)opt;
    for (int t = 0; t < opt; ++t) {
        uint64_t rpt = N / (uint64_t)opt;
        uint64_t a = (uint64_t)t * rpt;
        uint64_t b = (t == opt - 1) ? N : (uint64_t)(t + 1) * rpt;
        uint64_t local = (t == opt - 1) ? target - (uint64_t)(opt - 1) * per_thread : per_thread;
        if (local == 0) continue;
        uint64_t stride = umax(1, (b - a) / local);
        uint64_t got = 0;
        for (uint64_t i = a; i < b && got < local; i += stride) {
            emit(&s, i);
            ++got;
        }
    }
    return s.n;
}

/* ------------------------------------------------------------------------------------------------
 * R10 — region-per-worker strided sampling (DB.cpp:1880-1960 / 1962-2048)
 * ---------------------------------------------------------------------------------------------- */
int64_t aqo_idx_region_stride(uint64_t M, double pct, int T, uint64_t seed, const uint64_t* starts,
                              int reference_partition, uint64_t* out, int64_t cap) {
    sink s = {out, cap, 0};
    if (M == 0) return 0;
    if (T <= 0) return -1;
    double per_thread_pct = pct / (double)T; /* DB.cpp:1901 / 1982 */
    uint64_t region_size = M / (uint64_t)T, rem = M % (uint64_t)T;
    for (int t = 0; t < T; ++t) {
        uint64_t a, b;
        if (reference_partition) { /* DB.cpp:1926-1931: starts ignore the +1 remainders */
            a = (uint64_t)t * region_size;
            b = a + region_size + ((uint64_t)t < rem ? 1 : 0);
            if (a >= M) continue;
            if (b > M) b = M;
        } else { /* proper prefix partition */
            a = (uint64_t)(((__uint128_t)M * (uint64_t)t) / (uint64_t)T);
            b = (uint64_t)(((__uint128_t)M * (uint64_t)(t + 1)) / (uint64_t)T);
        }
        uint64_t total = b - a;
        uint64_t target = (uint64_t)((double)total * per_thread_pct / 100.0);
        if (target == 0) continue;
        uint64_t span = umin(total / 10, 100); /* start drawn from [a, a+span] inclusive, DB.cpp:1944 */
        uint64_t start = starts ? starts[t] : a + aqo_splitmix64_at(seed, (uint64_t)t) % (span + 1);
        uint64_t stride = total / target;
        if (stride == 0) stride = 1;
        uint64_t got = 0;
        for (uint64_t i = start; i < b && got < target; i += stride) {
            emit(&s, i);
            ++got;
        }
    }
    return s.n;
}

/* ------------------------------------------------------------------------------------------------
 * reductions
 * ---------------------------------------------------------------------------------------------- */
static void finish_moments(aqo_moments* m) {
    m->mean = m->n ? m->sum / (double)m->n : 0.0;
}

void aqo_moments_idx(const aqo_record* rows, const uint64_t* idx, int64_t n, int has_where, double wmin,
                     double wmax, aqo_moments* out) {
    memset(out, 0, sizeof *out);
    for (int64_t k = 0; k < n; ++k) {
        double x = rows[idx[k]].amount;
        if (has_where && !(x >= wmin && x <= wmax)) continue; /* inclusive both ends, DB.cpp:329 */
        out->n++;
        out->sum += x;
        out->sumsq += x * x;
    }
    finish_moments(out);
    for (int64_t k = 0; k < n; ++k) { /* second pass of CLI:280 */
        double x = rows[idx[k]].amount;
        if (has_where && !(x >= wmin && x <= wmax)) continue;
        out->m2 += (x - out->mean) * (x - out->mean);
    }
}

void aqo_moments_range(const aqo_record* rows, uint64_t lo, uint64_t hi, int has_where, double wmin,
                       double wmax, aqo_moments* out) {
    memset(out, 0, sizeof *out);
    for (uint64_t i = lo; i < hi; ++i) {
        double x = rows[i].amount;
        if (has_where && !(x >= wmin && x <= wmax)) continue; /* DB.cpp:269 */
        out->n++;
        out->sum += x;
        out->sumsq += x * x;
    }
    finish_moments(out);
    for (uint64_t i = lo; i < hi; ++i) {
        double x = rows[i].amount;
        if (has_where && !(x >= wmin && x <= wmax)) continue;
        out->m2 += (x - out->mean) * (x - out->mean);
    }
}

/* ------------------------------------------------------------------------------------------------
 * R5 / R7 / R11 — estimators, intervals, heuristics
 * ---------------------------------------------------------------------------------------------- */
double aqo_estimate_cli(int agg, uint64_t N, uint64_t n, double sum) {
    if (n == 0) return 0.0;
    switch (agg) {
        case AQO_SUM: return sum * ((double)N / (double)n); /* CLI:191-192 */
        case AQO_COUNT: return (double)N;                    /* CLI:196-197 */
        default: return sum / (double)n;                     /* CLI:193-200 */
    }
}

double aqo_estimate_cpp(int agg, uint64_t N, double pct, uint64_t n, double sum) {
    double scaled = sum * (100.0 / pct); /* DB.cpp:303 */
    switch (agg) {
        case AQO_SUM: return scaled;
        case AQO_AVG: return N ? scaled / (double)N : 0.0;                 /* DB.cpp:306-310 */
        default: return (double)(uint64_t)((double)n * (100.0 / pct));     /* DB.cpp:312-315 */
    }
}

double aqo_ci_cli(int agg, uint64_t N, uint64_t n, double m2, double estimate, double* lo, double* hi) {
    double var = m2 / (double)(n - 1);                       /* CLI:280 */
    double moe = 1.96 * pow(var, 0.5) / pow((double)n, 0.5); /* CLI:281-282 */
    double m = agg == AQO_SUM ? moe * ((double)N / (double)n) : moe; /* CLI:284-291 */
    *lo = estimate - m;
    *hi = estimate + m;
    return m;
}

double aqo_margin_moments(uint64_t n, double sum, double sumsq) {
    double c = (double)n;
    double var = (sumsq - (sum * sum / c)) / (c - 1.0); /* EXE:184 */
    return 1.96 * sqrt(var / c);                        /* EXE:187-190 */
}

double aqo_confidence_heuristic(double pct, uint64_t N) {
    double sample = (double)N * pct / 100.0;
    if (sample >= 1000) return 0.95;
    if (sample >= 500) return 0.90;
    if (sample >= 100) return 0.85;
    if (sample >= 50) return 0.80;
    return 0.70;
}

double aqo_error_to_percent(double e) {
    if (e <= 1.0) return 20.0;
    if (e <= 2.0) return 15.0;
    if (e <= 5.0) return 10.0;
    return 5.0;
}

double aqo_clt_zscore(double conf) { return conf >= 0.99 ? 2.576 : conf >= 0.95 ? 1.96 : 1.645; }

double aqo_clt_error_percent(uint64_t n, double mean, double var, double z) {
    double se = sqrt(var / (double)n); /* DB.cpp:954 */
    return (z * se / mean) * 100.0;    /* DB.cpp:955-956 */
}

int aqo_clt_fast_rule(uint64_t n, double mean, double var, double z, double e) {
    if (n < 30) return 0;                                             /* DB.cpp:936 */
    return aqo_clt_error_percent(n, mean, var, z) <= e && n >= 50;    /* DB.cpp:958 */
}

int aqo_clt_slow_rule(uint64_t n_slow, double mean_slow, uint64_t n_fast, double mean_fast, double e,
                      int base) {
    if (n_slow < 20 || n_fast < 30) return 0;  /* DB.cpp:993; a fast check must have published */
    if (!(mean_fast > 0)) return 0;            /* DB.cpp:1007 */
    double diff = fabs(mean_slow - mean_fast) / mean_fast;
    return diff <= e / 100.0 && n_fast >= (uint64_t)(base / 2); /* DB.cpp:1009-1011 */
}

/* ------------------------------------------------------------------------------------------------
 * R8 — CLT monitor
 * ---------------------------------------------------------------------------------------------- */
static uint64_t prog_count(uint64_t first, uint64_t end, uint64_t step) {
    return first < end ? (end - first + step - 1) / step : 0;
}

int aqo_clt_make_plan(uint64_t N, double pct, double conf, int check_interval, int T, aqo_clt_plan* p) {
    memset(p, 0, sizeof *p);
    p->base = target_of(N, pct);
    p->z = aqo_clt_zscore(conf);
    if (N == 0 || p->base <= 0) return 0; /* empty result, DB.cpp:894-898 */
    if (T <= 0 || T > AQO_MAX_WORKERS) return -1;
    int F = T / 2, S = T - F;
    if (check_interval / 2 == 0) return -1;    /* DB.cpp:993: modulo by zero */
    if (F > 0 && p->base / F == 0) return -1;  /* DB.cpp:927: division by zero */
    if (S > 0 && p->base / S == 0) return -1;  /* DB.cpp:985 */
    p->n_workers = T;
    p->n_fast = F;
    for (int t = 0; t < F; ++t) { /* DB.cpp:925-927 */
        uint64_t a = (N * (uint64_t)t) / (uint64_t)F, b = (N * (uint64_t)(t + 1)) / (uint64_t)F;
        int step = imax(3, (int)((b - a) / (uint64_t)(p->base / F)));
        aqo_clt_worker* w = &p->w[t];
        w->first = a; w->end = b; w->step = (uint64_t)step; w->is_fast = 1;
        w->group = t == 0 ? 0 : 1;
        w->count = prog_count(w->first, w->end, w->step);
    }
    for (int t = 0; t < S; ++t) { /* DB.cpp:983-990 */
        uint64_t a = (N * (uint64_t)t) / (uint64_t)S, b = (N * (uint64_t)(t + 1)) / (uint64_t)S;
        int step = imax(1, (int)((b - a) / (uint64_t)(p->base / S)));
        aqo_clt_worker* w = &p->w[F + t];
        w->first = a + (uint64_t)(step / 2); w->end = b; w->step = (uint64_t)step; w->is_fast = 0;
        w->group = 1;
        w->count = prog_count(w->first, w->end, w->step);
    }
    return 0;
}

static void acc(aqo_moments* m, double x) {
    m->n++;
    m->sum += x;
    m->sumsq += x * x;
}

void aqo_clt_round_partial(const aqo_record* rows_at_lo, uint64_t lo, uint64_t hi, const aqo_clt_plan* p,
                           uint64_t ord_begin, uint64_t ord_end, aqo_moments* fast, aqo_moments* slow) {
    memset(fast, 0, sizeof *fast);
    memset(slow, 0, sizeof *slow);
    for (int wi = 0; wi < p->n_workers; ++wi) {
        const aqo_clt_worker* w = &p->w[wi];
        uint64_t k1 = umin(ord_end, w->count);
        for (uint64_t k = ord_begin; k < k1; ++k) {
            uint64_t i = w->first + k * w->step;
            if (i < lo || i >= hi) continue;
            acc(w->group == 0 ? fast : slow, rows_at_lo[i - lo].amount);
        }
    }
}

/* exact two-pass m2 of everything sampled so far (what CLI:277-281 / DB.cpp:942-946 compute) */
static double m2_of(const double* v, uint64_t n, double mean) {
    double m2 = 0.0;
    for (uint64_t i = 0; i < n; ++i) m2 += (v[i] - mean) * (v[i] - mean);
    return m2;
}

int aqo_clt_run(const aqo_record* rows, uint64_t N, double pct, double conf, int check_interval, int T,
                double e, uint64_t R0, uint32_t growth, aqo_clt_result* res, uint64_t* idx_out,
                int64_t cap, int64_t* n_idx) {
    aqo_clt_plan plan;
    memset(res, 0, sizeof *res);
    if (n_idx) *n_idx = 0;
    int rc = aqo_clt_make_plan(N, pct, conf, check_interval, T, &plan);
    if (rc != 0) return rc;
    if (plan.base <= 0 || N == 0) return 0;
    if (R0 == 0 || growth == 0) return -1;
    sink s = {idx_out, cap, 0};

    uint64_t max_count = 0, total = 0;
    for (int i = 0; i < plan.n_workers; ++i) {
        if (plan.w[i].count > max_count) max_count = plan.w[i].count;
        total += plan.w[i].count;
    }
    /* values kept per group so the oracle can take the reference's exact two-pass variance */
    double* vf = (double*)malloc(sizeof(double) * (size_t)(total + (uint64_t)plan.base + 1));
    double* vs = (double*)malloc(sizeof(double) * (size_t)(total + 1));
    double* va = (double*)malloc(sizeof(double) * (size_t)(total + (uint64_t)plan.base + 1));
    if (!vf || !vs || !va) { free(vf); free(vs); free(va); return -1; }
    uint64_t nf = 0, ns = 0, na = 0;
    aqo_moments F = {0}, S = {0}, A = {0};

    uint64_t b = 0, R = R0;
    while (b < max_count) {
        uint64_t b1 = (R > max_count - b) ? max_count : b + R;
        for (int wi = 0; wi < plan.n_workers; ++wi) {
            const aqo_clt_worker* w = &plan.w[wi];
            uint64_t k1 = umin(b1, w->count);
            for (uint64_t k = b; k < k1; ++k) {
                uint64_t i = w->first + k * w->step;
                double x = rows[i].amount;
                emit(&s, i);
                acc(&A, x); va[na++] = x;
                if (w->group == 0) { acc(&F, x); vf[nf++] = x; } else { acc(&S, x); vs[ns++] = x; }
            }
        }
        res->rounds++;
        b = b1;
        R = (R > (UINT64_MAX / 4) / growth) ? UINT64_MAX / 4 : R * growth;
        /* rule A on the LEADER's own samples — fast worker 0 (DB.cpp:936-961: a fast thread judges the samples it took
         * itself; F = its moments, vf its values, two-pass variance as the reference takes it) */
        finish_moments(&A); finish_moments(&F); finish_moments(&S);
        if (F.n >= 30) {
            double var = m2_of(vf, nf, F.mean) / (double)(F.n - 1);
            if (aqo_clt_fast_rule(F.n, F.mean, var, plan.z, e)) { res->converged = 1; break; }
        }
        /* rule B: the others' mean cross-validates the leader's, once the leader holds base/2 rows (DB.cpp:993-1016) */
        if (aqo_clt_slow_rule(S.n, S.mean, F.n, F.mean, e, plan.base)) { res->converged = 2; break; }
    }
    A.m2 = m2_of(va, na, A.mean);
    F.m2 = m2_of(vf, nf, F.mean);
    S.m2 = m2_of(vs, ns, S.mean);
    res->all = A; res->fast = F; res->slow = S;

    /* top-up, DB.cpp:1031-1040 */
    aqo_moments Z = A;
    if (A.n < (uint64_t)(plan.base / 4)) {
        int additional = plan.base / 4;
        int step = imax(1, (int)(N / (uint64_t)additional));
        for (uint64_t i = 0; i < N && Z.n < (uint64_t)plan.base; i += (uint64_t)step) {
            double x = rows[i].amount;
            emit(&s, i);
            acc(&Z, x); va[na++] = x;
            res->topup++;
        }
    }
    finish_moments(&Z);
    Z.m2 = m2_of(va, na, Z.mean);
    res->final = Z;
    if (n_idx) *n_idx = s.n;
    free(vf); free(vs); free(va);
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * on-disk format, DB.cpp:665-711 (native endian, 24-byte header)
 * ---------------------------------------------------------------------------------------------- */
/* ---- GROUP BY (EXE:202-321) ---- */
static int64_t key_of(const aqo_record* r, int col) { return col == AQO_COL_REGION ? (int64_t)r->region : (int64_t)r->product_id; }

/* insertion into a small key-sorted table (the reference iterates SELECT DISTINCT; order is presentation only) */
static int64_t group_slot(int64_t key, int64_t* keys, uint64_t* n, double* sum, double* sumsq, int64_t* used, int64_t cap) {
    int64_t lo = 0, hi = *used < cap ? *used : cap;
    while (lo < hi) { int64_t mid = (lo + hi) / 2; if (keys[mid] < key) lo = mid + 1; else hi = mid; }
    int64_t filled = *used < cap ? *used : cap;
    if (lo < filled && keys[lo] == key) return lo;
    if (*used >= cap) { (*used)++; return -1; } /* counted, not stored */
    for (int64_t i = filled; i > lo; --i) { keys[i] = keys[i - 1]; n[i] = n[i - 1]; sum[i] = sum[i - 1]; sumsq[i] = sumsq[i - 1]; }
    keys[lo] = key; n[lo] = 0; sum[lo] = 0.0; sumsq[lo] = 0.0;
    (*used)++;
    return lo;
}

int64_t aqo_group_idx(const aqo_record* rows, const uint64_t* idx, int64_t n_idx, int group_col, int has_where, double wmin,
                      double wmax, int64_t* keys, uint64_t* n, double* sum, double* sumsq, int64_t cap) {
    int64_t used = 0;
    for (int64_t k = 0; k < n_idx; ++k) {
        const aqo_record* r = rows + idx[k];
        if (has_where && !(r->amount >= wmin && r->amount <= wmax)) continue;
        int64_t s = group_slot(key_of(r, group_col), keys, n, sum, sumsq, &used, cap);
        if (s < 0) continue;
        n[s] += 1; sum[s] += r->amount; sumsq[s] += r->amount * r->amount; /* EXE:236-238 */
    }
    return used;
}

int64_t aqo_group_rowid_mod(const aqo_record* rows, uint64_t N, int sample_percent, int group_col, int has_where,
                            double wmin, double wmax, int64_t* keys, uint64_t* n, double* sum, double* sumsq, int64_t cap) {
    int step = (sample_percent <= 0 || sample_percent >= 100) ? 0 : 100 / sample_percent; /* EXE:21-26 */
    int64_t used = 0;
    for (uint64_t i = 0; i < N; ++i) {
        if (step > 0 && (i + 1) % (uint64_t)step != 0) continue; /* rowid % step = 0, EXE:243 */
        const aqo_record* r = rows + i;
        if (has_where && !(r->amount >= wmin && r->amount <= wmax)) continue;
        int64_t s = group_slot(key_of(r, group_col), keys, n, sum, sumsq, &used, cap);
        if (s < 0) continue;
        n[s] += 1; sum[s] += r->amount; sumsq[s] += r->amount * r->amount;
    }
    return used;
}

void aqo_group_ci(int agg, uint64_t count, double sum, double sumsq, int sample_percent, int reference_sum, double* value,
                  double* lo, double* hi) {
    int step = (sample_percent <= 0 || sample_percent >= 100) ? 0 : 100 / sample_percent;
    double scale = 100.0 / (double)sample_percent;
    double c = (double)count;
    if (count < 2) { /* EXE:248-274: no interval */
        double v = agg == AQO_SUM ? sum : agg == AQO_AVG ? (count ? sum / c : 0.0) : c;
        if (step > 0 && agg != AQO_AVG) v *= scale;
        *value = *lo = *hi = v;
        return;
    }
    double mean = sum / c;                               /* EXE:277 */
    double var = (sumsq - (sum * sum / c)) / (c - 1.0);  /* EXE:280 */
    double margin = 1.96 * sqrt(var / c);                /* EXE:283-286 */
    double v = mean;
    if (agg == AQO_SUM) {                                /* EXE:289-296 */
        v = (reference_sum ? mean : sum) * scale;
        margin *= scale;
    } else if (agg == AQO_COUNT) {
        v = step > 0 ? c * scale : c;
        margin = 0.0;
    }
    *value = v; *lo = v - margin; *hi = v + margin;
}

int aqo_file_write(const char* path, const aqo_record* rows, uint64_t n, uint64_t height) {
    FILE* f = fopen(path, "wb");
    if (!f) return -1;
    uint64_t hdr[3] = {n, height, n};
    int ok = fwrite(hdr, sizeof hdr, 1, f) == 1 && (n == 0 || fwrite(rows, sizeof(aqo_record), n, f) == n);
    fclose(f);
    return ok ? 0 : -1;
}

int64_t aqo_file_count(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    uint64_t hdr[3];
    int ok = fread(hdr, sizeof hdr, 1, f) == 1;
    fclose(f);
    return ok ? (int64_t)hdr[2] : -1;
}

int64_t aqo_file_read(const char* path, aqo_record* rows, uint64_t first, uint64_t cap) {
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    uint64_t hdr[3];
    if (fread(hdr, sizeof hdr, 1, f) != 1) { fclose(f); return -1; }
    if (first > hdr[2]) first = hdr[2];
    uint64_t n = umin(cap, hdr[2] - first);
    if (fseek(f, (long)(24 + 32 * first), SEEK_SET) != 0) { fclose(f); return -1; }
    size_t got = n ? fread(rows, sizeof(aqo_record), n, f) : 0;
    fclose(f);
    return (int64_t)got;
}
