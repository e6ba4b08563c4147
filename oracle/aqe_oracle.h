/* oracle/aqe_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C) of the reference's sampled SUM/AVG/COUNT path, written from the
 * reference as a specification; every function cites the reference file:line it follows.
 * Abbreviations: DB.cpp = /root/reference/src/aqe_backend/core/custom_bplus_db.cpp,
 * DB.hpp = .../custom_bplus_db.hpp, SCH.cpp = .../custom_scheduler.cpp,
 * CLI = /root/reference/enhanced_aqe_cli.py, EXE = /root/reference/src/aqe_backend/executor.cpp.
 *
 * Who may use it: tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — as the checker
 * only.  The product (approximatequeryengine_amd/ + libaqe_hip.so) never links, loads or calls it.
 *
 * Parity pin: tests/golden/ref_*.json were produced by oracle/make_golden.py from the reference's own
 * C++ compiled here (oracle/_ref/libaqe_ref.so, see oracle/Makefile); tests/test_oracle_golden.py
 * checks this restatement against them (index sets bit-exact, sums <= 1e-12 rel).
 */
#ifndef AQE_ORACLE_H
#define AQE_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* DB.hpp:17-27 — 32-byte row, natural alignment, amount at byte 8. */
typedef struct {
    int64_t id;
    double amount;
    int32_t region;
    int32_t product_id;
    int64_t timestamp;
} aqo_record;

/* (n, sum x, sum x^2) in index order, plus the two-pass form the CLI uses (CLI:277-281). */
typedef struct {
    uint64_t n;
    double sum;
    double sumsq;
    double mean; /* sum / n                         */
    double m2;   /* sum (x - mean)^2, second pass   */
} aqo_moments;

/* ---- synthetic table: counter-based, shard-independent ---- */
uint64_t aqo_splitmix64_at(uint64_t seed, uint64_t i); /* i-th output of splitmix64(seed) */
double aqo_synth_amount(uint64_t seed, uint64_t i);    /* 1 + 999*u, u in [0,1) 53-bit     */
void aqo_synth_fill(aqo_record* rows, uint64_t first_row, uint64_t n, uint64_t seed);

/* ---- index-set generators.  out may be NULL (count only).  Return: number of indices the
 *      reference would return, or -1 where the reference itself divides by zero / loops forever. */
int64_t aqo_idx_memory_stride(uint64_t M, double pct, uint64_t stride_bytes, uint64_t* out, int64_t cap);
int64_t aqo_idx_address_arithmetic(uint64_t M, double pct, uint64_t* out, int64_t cap);
int64_t aqo_idx_random_start_stride(uint64_t M, double pct, uint64_t stride_bytes, uint64_t seed,
                                    const uint64_t* start_override, uint64_t* out, int64_t cap);
int64_t aqo_idx_random_pointer(uint64_t N, double pct, uint32_t seed, uint64_t* out, int64_t cap);
int64_t aqo_leaf_sizes(uint64_t N, uint32_t* sizes, int64_t cap);                               /* DB.cpp:164-240, 43-62 */
int64_t aqo_idx_direct_access(uint64_t N, double pct, uint64_t* out, int64_t cap);              /* DB.cpp:584-644 */
double aqo_uniform_real(uint32_t seed, double hi);                                              /* libstdc++ uniform_real_distribution over mt19937 */
int64_t aqo_idx_optimized_sequential(uint64_t N, double pct, uint32_t seed, uint64_t* out, int64_t cap); /* DB.cpp:366-428 */
int64_t aqo_idx_block(uint64_t N, double pct, uint64_t block_rows, uint64_t* out, int64_t cap);
int64_t aqo_idx_page(uint64_t N, double pct, uint64_t page_bytes, uint64_t* out, int64_t cap);
int64_t aqo_idx_parallel_block(uint64_t N, double pct, uint64_t block_rows, int T, uint64_t* out, int64_t cap);
int64_t aqo_idx_optimized_clt(uint64_t N, double pct, int T, uint64_t* out, int64_t cap);
/* data-dependent samplers ("next" rows of SURVEY §8f): need the rows */
int64_t aqo_idx_adaptive_block(const aqo_record* rows, uint64_t N, double pct, uint64_t min_block, uint64_t max_block,
                               double* zone_var_out, uint64_t* out, int64_t cap);
int64_t aqo_idx_stratified_block(const aqo_record* rows, uint64_t N, double pct, uint64_t B, int strata,
                                 uint64_t* out, int64_t cap);
int64_t aqo_idx_fast_pointer(uint64_t N, double pct, int step_size, uint64_t* out, int64_t cap);
int64_t aqo_idx_dual_pointer(uint64_t N, double pct, uint64_t* out, int64_t cap);
int64_t aqo_idx_parallel_pointer(uint64_t N, double pct, int T, uint64_t* out, int64_t cap);
/* region-per-worker strided sampler (DB.cpp:1880-2048).  starts==NULL: deterministic counter-based
 * start offsets keyed by (seed, t).  reference_partition!=0 reproduces the reference's own region
 * arithmetic (overlapping starts, DB.cpp:1926-1931) so a reference run can be replayed from the
 * start offsets it drew; 0 = proper prefix partition (what the product uses). */
int64_t aqo_idx_region_stride(uint64_t M, double pct, int T, uint64_t seed, const uint64_t* starts,
                              int reference_partition, uint64_t* out, int64_t cap);

/* ---- reductions over an index list ---- */
void aqo_moments_idx(const aqo_record* rows, const uint64_t* idx, int64_t n, int has_where,
                     double wmin, double wmax, aqo_moments* out);
void aqo_moments_range(const aqo_record* rows, uint64_t lo, uint64_t hi, int has_where, double wmin,
                       double wmax, aqo_moments* out);

/* ---- estimators and intervals ---- */
enum { AQO_SUM = 0, AQO_AVG = 1, AQO_COUNT = 2 };
double aqo_estimate_cli(int agg, uint64_t N, uint64_t n, double sum);             /* CLI:189-200 */
double aqo_estimate_cpp(int agg, uint64_t N, double pct, uint64_t n, double sum); /* DB.cpp:303-315 */
/* CLI:277-291: two-pass variance, 1.96, SUM margin scaled by N/n.  Returns margin of error. */
double aqo_ci_cli(int agg, uint64_t N, uint64_t n, double m2, double estimate, double* lo, double* hi);
/* EXE:180-199: moment form var=(Q-S^2/n)/(n-1).  Returns unscaled margin of the mean. */
double aqo_margin_moments(uint64_t n, double sum, double sumsq);
double aqo_confidence_heuristic(double pct, uint64_t N); /* SCH.cpp:296-305 */
double aqo_error_to_percent(double e);                   /* CLI:243-250      */
double aqo_clt_zscore(double conf);                      /* DB.cpp:911-912   */
/* DB.cpp:936-961 decision: inputs (n, mean, var) -> error percent; stop iff <= e and n >= 50. */
double aqo_clt_error_percent(uint64_t n, double mean, double var, double z);
int aqo_clt_fast_rule(uint64_t n, double mean, double var, double z, double e);
/* DB.cpp:1003-1016 cross-validation rule. */
int aqo_clt_slow_rule(uint64_t n_slow, double mean_slow, uint64_t n_fast, double mean_fast, double e,
                      int base);

/* ---- CLT monitor (DB.cpp:885-1043), round-synchronous restatement ---- */
#define AQO_MAX_WORKERS 256
typedef struct {
    uint64_t first; /* first sampled row      */
    uint64_t end;   /* exclusive range end    */
    uint64_t step;
    uint64_t count; /* samples in [first,end) */
    int is_fast;
    int group; /* 0: the LEADER (fast worker 0: its own samples decide the error rule, DB.cpp:936-961), 1: every other worker */
} aqo_clt_worker;

typedef struct {
    int base;      /* int(N*pct/100)                        */
    int n_workers; /* T                                     */
    int n_fast;    /* T/2                                   */
    double z;
    aqo_clt_worker w[AQO_MAX_WORKERS];
} aqo_clt_plan;

/* returns 0, or -1 for parameters on which the reference divides by zero (DB.cpp:927,985,993) */
int aqo_clt_make_plan(uint64_t N, double pct, double conf, int check_interval, int T, aqo_clt_plan* plan);

typedef struct {
    aqo_moments all, fast, slow; /* collected before top-up                      */
    aqo_moments final;           /* after the top-up of DB.cpp:1032-1040         */
    int converged;               /* 0 none, 1 rule A (fast), 2 rule B (slow)     */
    int rounds;                  /* rounds executed                              */
    uint64_t topup;              /* rows added by the top-up                     */
} aqo_clt_result;

/* Round r takes ordinals [b_r, b_r + R0*growth^r) of every worker's progression, then evaluates the
 * rules on the pooled moments.  e<=0 on non-constant data never converges: the sample multiset is
 * then exactly the reference's (2*base rows).  idx_out (optional) receives every sampled row index,
 * top-up included, in (round, worker, ordinal) order. */
int aqo_clt_run(const aqo_record* rows, uint64_t N, double pct, double conf, int check_interval, int T,
                double max_error_percent, uint64_t R0, uint32_t growth, aqo_clt_result* res,
                uint64_t* idx_out, int64_t cap, int64_t* n_idx);

/* Partial (per-shard) moments of one round: rows points at row `lo`; only indices in [lo,hi) count. */
void aqo_clt_round_partial(const aqo_record* rows_at_lo, uint64_t lo, uint64_t hi,
                           const aqo_clt_plan* plan, uint64_t ord_begin, uint64_t ord_end,
                           aqo_moments* fast_sums, aqo_moments* slow_sums);

/* ---- GROUP BY with a per-group interval: the SQLite executor's semantics (EXE:202-321) on the same rows ----
 * Sample = rows whose rowid (= row index + 1) satisfies rowid % step == 0, step = 100 / sample_percent
 * (integer division; EXE:21-26; sample_percent <= 0 or >= 100: every row).  Per distinct key of the group
 * column among the SAMPLED rows that pass WHERE: COUNT, SUM, SUM(x*x) in rowid order (EXE:236-243).
 * keys ascending.  Returns the number of groups (may exceed cap; only cap are written). */
enum { AQO_COL_REGION = 1, AQO_COL_PRODUCT = 2 };
int64_t aqo_group_rowid_mod(const aqo_record* rows, uint64_t N, int sample_percent, int group_col, int has_where,
                            double wmin, double wmax, int64_t* keys, uint64_t* n, double* sum, double* sumsq, int64_t cap);
/* the same grouping over an explicit index list (any sampler), index order */
int64_t aqo_group_idx(const aqo_record* rows, const uint64_t* idx, int64_t n_idx, int group_col, int has_where, double wmin,
                      double wmax, int64_t* keys, uint64_t* n, double* sum, double* sumsq, int64_t cap);
/* EXE:247-302: (count, sum, sumsq) of one group -> value and interval.  count < 2: no interval, value = the
 * aggregate of the sampled rows, scaled by 100/pct unless AVG (EXE:248-274).  Otherwise mean +/- 1.96 sqrt(var/n);
 * for SUM the reference scales the MEAN by 100/pct (EXE:291-296, reproduced when reference_sum != 0); with
 * reference_sum == 0 the value is sum * 100/pct with the same scaled margin (what the product reports). */
void aqo_group_ci(int agg, uint64_t count, double sum, double sumsq, int sample_percent, int reference_sum, double* value,
                  double* lo, double* hi);

/* ---- on-disk format (DB.cpp:665-711): size_t total | size_t height | size_t count | rows ---- */
int aqo_file_write(const char* path, const aqo_record* rows, uint64_t n, uint64_t height);
int64_t aqo_file_count(const char* path);
int64_t aqo_file_read(const char* path, aqo_record* rows, uint64_t first, uint64_t cap);

/* mt19937 raw stream (for pinning against std::mt19937 golden values) */
void aqo_mt19937_stream(uint32_t seed, uint32_t* out, int n);

#ifdef __cplusplus
}
#endif
#endif
