/* include/aqe_hip.h — C ABI of libaqe_hip.so, the MI355X (gfx950) execution path for the
 * reference's sampled SUM/AVG/COUNT reducer with CLT confidence interval.
 *
 * What it replaces.  The reference has no C ABI: its boundary is the pybind11 module `aqe_backend`
 * (/root/reference/src/aqe_backend/bindings/bindings.cpp:10-137) whose class CustomBPlusDB *is* the
 * operator API.  Each entry point below names the reference interface it stands in for; the Python
 * mirror (approximatequeryengine_amd/aqe_backend.py) and INTEGRATION.md show the binding.
 *   DB.cpp = src/aqe_backend/core/custom_bplus_db.cpp, DB.hpp = .../custom_bplus_db.hpp,
 *   SCH.cpp = .../custom_scheduler.cpp, BIND = src/aqe_backend/bindings/bindings.cpp,
 *   CLI = enhanced_aqe_cli.py, EXE = src/aqe_backend/executor.cpp.
 *
 * Conventions.  Plain C, POD structs, caller-allocated outputs, no C++/torch types.  Every function
 * returns an aqe_status (0 = ok, negative = error; aqe_last_error() has the text).  A context is
 * bound to one GPU and must be used from one host thread at a time.  There is no CPU fallback: with
 * no usable HIP device aqe_create fails with AQE_ERR_NO_DEVICE.
 *
 * Data model.  A context holds one *shard*: rows [shard_lo, shard_lo + local_rows) of a table of
 * global_rows rows in flat leaf order (the reference's `cached_records_`, DB.hpp:158-159).  Rows are
 * staged once into HBM as a structure of arrays: the `amount` column (f64, the only column the
 * reducers read) plus, optionally, the 32-byte AoS rows for the record-returning samplers.  All
 * sampler arithmetic is on GLOBAL row indices, so the union over shards of what each shard samples
 * is bit-identical to the single-device index set.
 */
#ifndef AQE_HIP_H
#define AQE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AQE_API __attribute__((visibility("default")))
#define AQE_ABI_VERSION 2

typedef struct aqe_ctx aqe_ctx;   /* one GPU, one shard                              */
typedef struct aqe_plan aqe_plan; /* a planned query: families, rounds, device state */

typedef enum aqe_status {
    AQE_OK = 0,
    AQE_ERR_INVALID = -1,   /* bad argument, or parameters on which the reference divides by zero */
    AQE_ERR_HIP = -2,       /* a HIP call failed                                                  */
    AQE_ERR_NO_DEVICE = -3, /* no usable gfx950 device / HIP runtime                              */
    AQE_ERR_NO_TABLE = -4,  /* nothing staged                                                     */
    AQE_ERR_IO = -5,        /* file missing / malformed                                           */
    AQE_ERR_CAPACITY = -6,  /* caller buffer too small                                            */
    AQE_ERR_UNSUPPORTED = -7
} aqe_status;

/* DB.hpp:17-27 — the reference's 32-byte row, amount at byte 8. */
typedef struct aqe_record {
    int64_t id;
    double amount;
    int32_t region;
    int32_t product_id;
    int64_t timestamp;
} aqe_record;

/* Sampler selection; the number is the reference method it reproduces. */
typedef enum aqe_method {
    AQE_M_EXACT = 0,              /* sum_amount / sum_amount_where          DB.cpp:242-274   */
    AQE_M_MEMORY_STRIDE = 1,      /* memory_stride_sample                   DB.cpp:1526-1603 */
    AQE_M_ADDRESS_ARITHMETIC = 2, /* optimized_address_arithmetic_sample    DB.cpp:1667-1703 */
    AQE_M_RANDOM_POINTER = 3,     /* random_pointer_sample (mt19937+Lemire) DB.cpp:856-882   */
    AQE_M_BLOCK = 4,              /* block_sample                           DB.cpp:1151-1181 */
    AQE_M_PAGE = 5,               /* page_sample                            DB.cpp:1183-1216 */
    AQE_M_PARALLEL_BLOCK = 6,     /* parallel_block_sample                  DB.cpp:1218-1271 */
    AQE_M_OPTIMIZED_CLT = 7,      /* optimized_clt_sample                   DB.cpp:1046-1147 */
    AQE_M_CLT_DUAL_POINTER = 8,   /* clt_validated_dual_pointer_sample      DB.cpp:885-1043  */
    AQE_M_FAST_POINTER = 9,       /* fast_pointer_sample                    DB.cpp:737-758   */
    AQE_M_SLOW_POINTER = 10,      /* slow_pointer_sample                    DB.cpp:760-780   */
    AQE_M_DUAL_POINTER = 11,      /* dual_pointer_sample                    DB.cpp:782-813   */
    AQE_M_PARALLEL_POINTER = 12,  /* parallel_pointer_sample                DB.cpp:815-854   */
    AQE_M_REGION_STRIDE = 13,     /* multithreaded_memory_stride_sample / fast_aggregated_memory_stride_sum,
                                     DB.cpp:1880-2048, with a seeded counter-based start per region */
    AQE_M_RANDOM_START_STRIDE = 14, /* random_start_memory_stride_sample, DB.cpp:1838-1878, seeded start in [0, stride) */
    AQE_M_ADAPTIVE_BLOCK = 15,    /* adaptive_block_sample, DB.cpp:1273-1329: block size from per-zone variance (needs a
                                     full-table moments pre-pass on the device, cached per table)            */
    AQE_M_ROWID_MOD = 17,         /* the SQLite executor's sampler: rows with rowid % (100 / int(sample_percent)) == 0,
                                     rowid = row + 1 (executor.cpp:21-26, 36-41); sample_percent >= 100: every row */
    AQE_M_STRATIFIED_BLOCK = 16,  /* stratified_block_sample, DB.cpp:1331-1379: blocks of the amount-SORTED table (needs a
                                     device sort of the column, cached per table); num_threads = strata_count */
    AQE_M_RANDOM_DEVICE = 18,     /* a simple random sample WITHOUT replacement of int(N pct/100) rows, drawn on the device:
                                     row = P_seed(k), k = 0 .. target-1, where P_seed is a keyed bijection of [0, N) (multiply /
                                     xor-shift rounds on ceil(log2 N) bits, cycle-walked into [0, N)).  Counter-based: no host
                                     index list (RANDOM_POINTER draws mt19937 + Lemire on the host, 4.7 ns per index), any grid,
                                     any sharding, replayable from (seed, N).  It stands in for the reference's random_device-
                                     seeded samplers (sample_records, DB.cpp:345-363: shuffle all rows, take a prefix — hence
                                     parallel_{sum,avg,count}[_where]_sample, DB.cpp:276-343), which admit statistical parity
                                     only; RANDOM_POINTER stays the bit-exact restatement of random_pointer_sample(seed) */
    AQE_M_DIRECT_ACCESS = 19,     /* direct_access_sample, DB.cpp:584-644 — what the reference CLI takes for 10 k < N <= 50 k rows
                                     (CLI:181-183): ~10 % of the B+ tree's leaves at a fixed node step, evenly spaced records in
                                     each.  The leaves are those the reference builds from ascending inserts (insert_batch sorts
                                     by id; load_from_file): 127 rows each, the last 128 ... 254.  Deterministic; a leaf visited
                                     twice gives its rows twice, as in the reference.  An explicit row list (k_indexed) */
    AQE_M_OPTIMIZED_SEQUENTIAL = 20 /* optimized_sequential_sample, DB.cpp:366-428 — the CLI's sampler for N <= 10 k rows
                                     (CLI:184-186): one row whenever the running count reaches the next sample point, which
                                     advances by 100 / pct from a random start in [0, 100 / pct).  The reference seeds the start
                                     from std::random_device (statistical parity only); here `seed` feeds mt19937 the way
                                     libstdc++'s uniform_real_distribution would read it */
} aqe_method;

typedef enum aqe_agg { AQE_SUM = 0, AQE_AVG = 1, AQE_COUNT = 2 } aqe_agg;

/* How (n, S) become the reported value. */
typedef enum aqe_convention {
    AQE_EST_CLI = 0, /* CLI:189-200   SUM = S*(N/n), AVG = S/n, COUNT = N                         */
    AQE_EST_CPP = 1, /* DB.cpp:303-315 SUM = S*(100/pct), AVG = SUM/N, COUNT = size_t(n*100/pct)   */
    AQE_EST_RAW = 2  /* DB.cpp:2046   unscaled sample sum (fast_aggregated_memory_stride_sum)     */
} aqe_convention;

typedef struct aqe_query {
    int32_t method;           /* aqe_method                                                      */
    int32_t agg;              /* aqe_agg                                                         */
    int32_t convention;       /* aqe_convention                                                  */
    int32_t num_threads;      /* T of the parallel / CLT samplers (reference default 4)          */
    double sample_percent;    /* pct, in percent                                                 */
    uint64_t stride_bytes;    /* memory_stride_sample: 0 = auto (DB.cpp:1549-1556)               */
    uint64_t block_size;      /* block rows (BLOCK/PARALLEL_BLOCK/STRATIFIED), page bytes (PAGE), or
                                 min_block_size (ADAPTIVE_BLOCK)                                  */
    uint64_t seed;            /* RANDOM_POINTER (low 32 bits), REGION_STRIDE                     */
    int32_t step_size;        /* FAST_POINTER multiplier (reference default 2)                   */
    int32_t check_interval;   /* CLT (reference default 10)                                      */
    double confidence_level;  /* CLT: picks z = 2.576 / 1.96 / 1.645 (DB.cpp:911-912)            */
    double max_error_percent; /* CLT: e, in percent (DB.cpp:958)                                 */
    int32_t has_where;        /* 1: keep min <= amount <= max, both inclusive (DB.cpp:329)        */
    int32_t reserved0;
    double where_min, where_max;
    uint64_t clt_round0;      /* CLT: samples per worker in round 0; 0 = check_interval          */
    uint32_t clt_growth;      /* CLT: round r takes clt_round0*growth^r per worker; 0/1 = fixed   */
    uint32_t flags;           /* AQE_Q_*                                                         */
    uint64_t visible_rows;    /* M of the cached samplers; 0 = global_rows (the
                                 reference's stale-cache quirk DB.cpp:188-191 is not reproduced)  */
    uint64_t block_size_max;  /* ADAPTIVE_BLOCK: max_block_size (reference default 2000)          */
    uint64_t row_lo, row_hi;  /* row_hi > row_lo: the sampler runs over rows [row_lo, row_hi) only, as if they
                                 were the whole table (key-range pruning: see aqe_key_range_rows); N in the
                                 estimators is then row_hi - row_lo                                */
} aqe_query;

#define AQE_Q_NO_TOPUP 1u   /* CLT: skip the systematic top-up of DB.cpp:1031-1040 */
#define AQE_Q_NO_PERSIST 2u /* run every round as its own launch even on one GPU (same results) */
#define AQE_Q_NO_LAYOUT 8u /* CLT: sweep the sampled rows where they lie in the column; by default the column is kept a
                              second time in stride-major order per pointer step in use, where a pointer's rows are
                              contiguous (same rows, same answer, a fraction of the memory traffic) */
#define AQE_Q_SHARE_GPU 16u /* this query will run beside others (several plans in flight on different streams): its
                               single-launch sweep takes half the compute units, so that two fit the chip side by side */
#define AQE_Q_NO_LEAN 32u /* single-launch form: keep the persistent sweep with its monitor wave (k_sweep_persist) where the
                             plan would qualify for the lean launch that judges every round once, at the end (k_sweep_lean:
                             small sweeps, every family a plain run of rows).  Same decision rule on the same partial
                             moments; the sums are taken in another order */
#define AQE_Q_FORCE_LEAN 64u /* a single-round sampler (exact scan, strided sample through a view, blocks) takes the lean
                               launch at any size; by default only sweeps of >= 1536 tiles (12 MB) do, smaller ones stay
                               with k_round.  Same rows, same answer */
#define AQE_Q_FORCE_PERSIST 4u /* take the single-launch form whenever the plan has one, also where the query is
                                  predicted to stop early (by default such plans are launched round by round) */

/* Everything a caller of the reference computes from a sample, produced on the device. */
typedef struct aqe_result {
    double value;      /* estimate under query.convention                                   */
    double ci_lower;   /* value -/+ margin; CLI:277-291 (1.96, two-pass variance)            */
    double ci_upper;
    double margin;     /* half-width actually applied to `value`                             */
    double sum;        /* S  = sum of sampled amounts that pass WHERE                        */
    double sumsq;      /* Q  = sum of squares                                                */
    double mean;       /* S / n                                                              */
    double m2;         /* sum (x - mean)^2                                                   */
    uint64_t n;        /* samples folded into (S, Q) (pass WHERE)                            */
    uint64_t visited;  /* samples drawn (== n without WHERE)                                 */
    uint64_t topup;    /* CLT: rows added by the top-up                                      */
    int32_t converged; /* CLT: 0 no, 1 error rule on the leader's own samples (DB.cpp:958), 2 cross-validation of the others' mean
                          against the leader's (DB.cpp:1009) */
    int32_t rounds;    /* CLT: rounds folded before the stop                                 */
    double kernel_ms;  /* aqe_reduce / timed executions: device time of the query by the device's own 100 MHz clock,
                          from its first launch starting (its first workgroup, for a single-launch form) to the result
                          being written; a replayed graph of launches is timed by two events around it instead */
    uint64_t bytes_algorithmic; /* 8 B per visited sample (SoA amount column)                 */
    int32_t device_status; /* 0 ok; nonzero: the device-side round protocol reported an error   */
    int32_t topup_pending; /* batched multi-GPU form only: 1 = the top-up (DB.cpp:1031-1040) is due and has
                              not been applied; run it as one more step (see aqe_plan_enqueue_replay) */
} aqe_result;

/* One arithmetic family of sampled rows: row(o) = row0 + (o / seg_len) * pitch + (o % seg_len) * step
 * for ordinals o in [ord_lo, ord_hi).  Every deterministic sampler is a short list of these.
 * A PAIR family carries a second pointer with the same step over the same rows — the reference's
 * fast and slow pointer of one region (DB.cpp:925-927 / 983-987) — so one sweep serves both:
 * row_b(o) = row0_b + o * step for o in [ord_lo_b, ord_hi_b), folded into group 1 (the first pointer into `group`). */
typedef struct aqe_family {
    uint64_t row0, pitch, seg_len, step;
    uint64_t ord_lo, ord_hi;
    uint64_t row0_b, ord_lo_b, ord_hi_b; /* AQE_F_PAIR only */
    uint32_t group; /* 0 = default; CLT: the LEADER (fast worker 0, whose own statistics decide the error rule,
                       DB.cpp:936-961), 1 = every other worker (the other fast pointers and the slow ones) */
    uint32_t flags; /* AQE_F_* */
} aqe_family;
#define AQE_F_TOPUP 1u /* ord_hi is further limited on the device to base - collected */
#define AQE_F_PAIR 2u  /* second pointer present (single segment families only) */

typedef struct aqe_table_info {
    uint64_t global_rows, shard_lo, local_rows;
    double shift;        /* c of the shifted moments (mean of the table's first <=1024 rows unless set) */
    int32_t has_aos;     /* 32-byte rows resident (record-returning samplers available)      */
    int32_t device_id;
    uint64_t hbm_bytes;  /* bytes this context holds in HBM (table + views + key columns + sort)  */
    uint64_t view_bytes; /* ... of which stride-major views of the column (and of key columns): one per pointer step in
                            use, at most 8 per table; past that the least recently used view no live plan holds is evicted */
    uint32_t n_views;
    uint32_t view_evictions; /* views dropped to make room since the table was staged                        */
    uint32_t view_fallbacks; /* plans that wanted a view while all 8 were held by live plans: swept in place  */
    uint32_t reserved;
} aqe_table_info;

/* ---- lifecycle ------------------------------------------------------------------------------ */
AQE_API int aqe_abi_version(void);
/* replaces: CustomBPlusDB() (BIND:42-43, DB.cpp:123-129).  device_id: HIP ordinal. */
AQE_API int aqe_create(int device_id, aqe_ctx** out);
AQE_API void aqe_destroy(aqe_ctx* ctx);
AQE_API const char* aqe_last_error(const aqe_ctx* ctx); /* ctx may be NULL: last create() error */
AQE_API const char* aqe_status_string(int status);

/* ---- staging: replaces insert_record/insert_batch + collect_leaf_records ("mmap") ------------ */
#define AQE_STAGE_KEEP_AOS 1u /* also keep the 32-byte rows in HBM (needed by aqe_gather)     */
/* Host rows (any pageable/mmap'd memory) -> HBM through pinned double buffers.  rows = this shard's
 * rows [shard_lo, shard_lo+n_local) of a table of n_global rows.   DB.cpp:164-194, 715-735. */
AQE_API int aqe_stage_records(aqe_ctx* ctx, const void* aos32, uint64_t n_local, uint64_t shard_lo,
                              uint64_t n_global, uint32_t flags);
/* The reference's file format (24-byte header + AoS, DB.cpp:665-711): mmap + stage rows
 * [shard_lo, shard_lo+n_local) (n_local = 0: to the end).  Replaces open_database/load_from_file. */
AQE_API int aqe_stage_file(aqe_ctx* ctx, const char* path, uint64_t shard_lo, uint64_t n_local,
                           uint32_t flags);
AQE_API int aqe_file_rows(const char* path, uint64_t* n_rows); /* header only */
/* Where the time of the most recent aqe_stage_records / aqe_stage_file of this context went (wall clock, milliseconds).
 * Rows travel host -> ring of pinned buffers (filled by a pool of host threads: memcpy from host rows, pread from a file)
 * -> hipMemcpyAsync -> HBM; the ring stays with the context, so only the first staging pays pinned_alloc_ms. */
typedef struct aqe_stage_stats {
    double total_ms;         /* the whole call                                                          */
    double device_alloc_ms;  /* hipMalloc of the column (and of the rows with AQE_STAGE_KEEP_AOS)       */
    double pinned_alloc_ms;  /* allocating the pinned ring (0 when the context already had it)          */
    double fill_ms;          /* host threads filling pinned buffers (wall time, summed over chunks)     */
    double wait_ms;          /* the host waiting for the copy engine: a buffer still draining, the final drain */
    uint64_t host_bytes;     /* bytes read from host rows / the file                                    */
    uint64_t link_bytes;     /* bytes sent over PCIe                                                    */
    uint32_t chunks, fill_threads;
} aqe_stage_stats;
AQE_API int aqe_last_stage_stats(const aqe_ctx* ctx, aqe_stage_stats* out);
/* Writes the staged shard back in the reference's format (save_to_file, DB.cpp:665-683). */
AQE_API int aqe_save_file(aqe_ctx* ctx, const char* path);
/* Synthetic `sales` shard generated in HBM: id=i+1, amount=1+999*u(splitmix64(seed,i)). */
AQE_API int aqe_generate_synthetic(aqe_ctx* ctx, uint64_t n_local, uint64_t shard_lo, uint64_t n_global,
                                   uint64_t seed, uint32_t flags);
/* Adopt caller-owned device memory (e.g. a torch tensor): f64 amount column, optional AoS rows. */
AQE_API int aqe_attach_device(aqe_ctx* ctx, const double* dev_amount, const void* dev_aos32,
                              uint64_t n_local, uint64_t shard_lo, uint64_t n_global, double shift);
AQE_API int aqe_set_shift(aqe_ctx* ctx, double shift); /* all shards of one table must agree */
AQE_API int aqe_table_info_get(const aqe_ctx* ctx, aqe_table_info* out);
/* B+-tree key bounds -> row interval.  Rows are in ascending-id leaf order (DB.cpp:715-735), so
 * `WHERE id BETWEEN id_min AND id_max` is the row window [*row_lo, *row_hi) — what the reference's declared but
 * never defined BPlusTreeNode::search_range (DB.hpp:45) would have pruned to.  Needs the whole table in this
 * context, and either dense ids (id = first_id + row, detected at staging) or the rows resident (KEEP_AOS). */
AQE_API int aqe_key_range_rows(aqe_ctx* ctx, int64_t id_min, int64_t id_max, uint64_t* row_lo, uint64_t* row_hi);
/* The same (BPlusTreeNode::search_range, DB.hpp:45; leaf order DB.cpp:715-735) for a SHARDED table: how many of THIS context's rows have id < id_min (*n_below) and id <= id_max (*n_upto).  Ids
 * ascend over the whole table, so the global window is [sum of n_below, sum of n_upto) over the ranks: one all-reduce SUM of
 * two numbers, then aqe_query.row_lo / row_hi as above on every rank. */
AQE_API int aqe_key_range_counts(aqe_ctx* ctx, int64_t id_min, int64_t id_max, uint64_t* n_below, uint64_t* n_upto);
AQE_API int aqe_release_table(aqe_ctx* ctx);

/* ---- the variance-aware samplers over a SHARDED table (SURVEY 8e "what does not shard") ------------------------------
 * adaptive_block_sample (DB.cpp:1273-1329) sizes its blocks from the population variance of ten zones of the whole table
 * (DB.cpp:1291-1308), stratified_block_sample (DB.cpp:1331-1379) takes blocks of the table SORTED by amount (DB.cpp:1342-1345):
 * both need something global before a shard can plan.  A context that holds the whole table does that by itself
 * (aqe_reduce / aqe_plan_create); the ranks of a sharded table exchange it with one all-reduce each:
 *   adaptive    aqe_zone_moments on every rank -> all-reduce SUM of the 30 doubles -> var_z = Q_z/n_z - (S_z/n_z)^2
 *               (the reference's expression) -> aqe_set_zone_variances on every rank -> aqe_plan_create plans the same
 *               blocks everywhere, clipped to the rank's rows.
 *   stratified  positions in the GLOBAL sorted order map to positions in each rank's own sorted column through the value
 *               found there: aqe_sorted_counts answers "how many of my rows are < v / <= v" for a list of values (the
 *               host bisects on v with an all-reduce of the counts per step), the global blocks become runs of the local
 *               sorted column, and aqe_plan_create_families plans a query over exactly those runs.
 * distributed.py (sharded_adaptive_plan, sharded_stratified_plan) is the host side of both. */
/* out30[3 z + {0,1,2}] = (rows, sum of amounts, sum of squared amounts) of the rows of zone z = global rows
 * [z * (N / 10), min((z + 1) * (N / 10), N)) that this context holds (zeros where it holds none). */
AQE_API int aqe_zone_moments(aqe_ctx* ctx, double* out30);
/* The ten zone variances of the whole table, as agreed among the ranks; kept until the table changes. */
AQE_API int aqe_set_zone_variances(aqe_ctx* ctx, const double* var10);
/* For each values[i]: how many rows of this context have amount < values[i] (n_less) and <= values[i] (n_less_equal).
 * Sorts the context's column on first use (kept until the table changes). */
AQE_API int aqe_sorted_counts(aqe_ctx* ctx, const double* values, uint32_t n, uint64_t* n_less, uint64_t* n_less_equal);

/* Device scratch for hosts that do not link the HIP runtime themselves (the moment vectors and total buffers of the
 * multi-GPU entry points live in device memory): plain hipMalloc / hipFree / synchronous copies to and from the host. */
AQE_API int aqe_device_malloc(aqe_ctx* ctx, size_t bytes, void** out);
AQE_API int aqe_device_free(aqe_ctx* ctx, void* dev_ptr);
AQE_API int aqe_device_read(aqe_ctx* ctx, void* host_dst, const void* dev_src, size_t bytes, void* stream);
AQE_API int aqe_device_write(aqe_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes, void* stream);

/* ---- host-side planning (no GPU needed) ------------------------------------------------------ */
AQE_API void aqe_query_defaults(aqe_query* q); /* reference defaults of BIND:56-101 */
/* Families of `q` over a table of n_global rows, clipped to rows [shard_lo, shard_hi).  For the CLT
 * sampler `round` selects the round (families of the top-up come last, flagged AQE_F_TOPUP, when
 * round == rounds).  Returns the number of families in *n_out (fams may be NULL to count).
 * RANDOM_POINTER has no families: use aqe_plan_random_indices. */
AQE_API int aqe_plan_families(const aqe_query* q, uint64_t n_global, uint64_t shard_lo, uint64_t shard_hi,
                              uint32_t round, aqe_family* fams, uint32_t cap, uint32_t* n_out,
                              uint32_t* rounds_out, uint64_t* samples_out);
/* adaptive_block_sample is data dependent: its families follow from the ten zone variances (population
 * variance of the amounts of rows [z*N/10, (z+1)*N/10)), which aqe_reduce obtains with a device pre-pass.
 * This host-side entry plans it from given variances (tests, external planners). */
AQE_API int aqe_plan_adaptive_families(const aqe_query* q, uint64_t n_global, const double* zone_var10,
                                       aqe_family* fams, uint32_t cap, uint32_t* n_out, uint64_t* samples_out);
/* Ascending unique indices of random_pointer_sample(pct, seed) that fall in [shard_lo, shard_hi). */
AQE_API int aqe_plan_random_indices(uint64_t n_global, double pct, uint32_t seed, uint64_t shard_lo,
                                    uint64_t shard_hi, uint64_t* out, uint64_t cap, uint64_t* n_out);
/* The explicit row list of a sampler that has no families — RANDOM_POINTER, DIRECT_ACCESS, OPTIMIZED_SEQUENTIAL — in the
 * reference's order, restricted to [shard_lo, shard_hi) (a row window of the query applies). */
AQE_API int aqe_plan_row_list(const aqe_query* q, uint64_t n_global, uint64_t shard_lo, uint64_t shard_hi, uint64_t* out,
                              uint64_t cap, uint64_t* n_out);
/* WHERE-range extraction of the façade (SCH.cpp:277-294): returns 1 and fills lo/hi, 0 if none. */
AQE_API int aqe_parse_where(const char* query, double* lo, double* hi);
AQE_API double aqe_confidence_heuristic(double sample_percent, uint64_t total_records); /* SCH.cpp:296-305 */
AQE_API double aqe_error_to_sample_percent(double error_percent);                       /* CLI:243-250 */

/* ---- the hot path ---------------------------------------------------------------------------- */
/* One complete approximate aggregate on this context's GPU (shard must be the whole table):
 * sample -> (n, S, Q) -> [CLT rounds with device-side should_stop] -> estimate + interval.
 * Replaces: <sampler>(...) + CLI:189-200/262-291, fast_aggregated_memory_stride_sum (BIND:98-99),
 * parallel_{sum,avg,count}[_where]_sample (DB.cpp:276-343), sum_amount[_where] (BIND:48-49). */
AQE_API int aqe_reduce(aqe_ctx* ctx, const aqe_query* q, aqe_result* out);

/* Record-returning form of the same samplers (BIND:50-101): rows in the reference's order
 * (CLT: round-major order; compare as a multiset).  Needs AQE_STAGE_KEEP_AOS. */
AQE_API int aqe_gather(aqe_ctx* ctx, const aqe_query* q, void* out_aos32, uint64_t cap, uint64_t* n_out);

/* ---- GROUP BY with a per-group interval -------------------------------------------------------
 * The same sampled sweep with one (n, S, Q) bin per key of `group_column` (region or product_id), then estimate
 * and interval per group.  Replaces execute_query_groupby_with_ci (executor.cpp:202-321, the reference's SQLite
 * path; use AQE_M_ROWID_MOD for its `rowid % step = 0` sample, any other single-round family sampler works too).
 * Per group: mean = S/n, var = (Q - S^2/n)/(n-1), half-width 1.96 sqrt(var/n) (n >= 2, else no interval);
 * AVG reports the mean; SUM reports S * 100/pct with the half-width scaled by 100/pct as the reference does
 * (the reference scales the MEAN and calls it the sum, executor.cpp:289-296: that defect is not reproduced —
 * `mean` is returned beside it); COUNT reports n * 100/pct without an interval.  query.convention is ignored.
 * Groups come back in ascending key order; only keys with at least one sampled row are listed.
 * Needs the key columns: stage with AQE_STAGE_KEEP_AOS, or a table made by aqe_generate_synthetic. */
#define AQE_GROUP_REGION 1
#define AQE_GROUP_PRODUCT 2
typedef struct aqe_group_result {
    int64_t key;
    uint64_t n;       /* sampled rows of the group that pass WHERE */
    uint64_t visited; /* sampled rows of the group                 */
    double sum, sumsq, mean;
    double value, ci_lower, ci_upper;
} aqe_group_result;
AQE_API int aqe_reduce_grouped(aqe_ctx* ctx, const aqe_query* q, int group_column, aqe_group_result* out, uint32_t cap,
                               uint32_t* n_groups);
/* Multi-GPU form: the bins are additive, so every rank bins the part of the sample that falls in its shard over
 * the SAME key range and one all-reduce SUM merges them:
 *     aqe_group_key_range(ctx, column, &kmin, &kmax)        this shard's keys (empty shard: INT32_MAX, INT32_MIN);
 *                                                            all-reduce MIN / MAX them, nbins = kmax - kmin + 1 <= 1024
 *     aqe_grouped_enqueue_bins(ctx, q, column, kmin, nbins, dev_bins, stream)    nbins x 4 doubles per rank:
 *                                                            {n, S - c n, Q (shifted), visited} per key
 *     <all-reduce SUM of nbins * 4 doubles on `stream`>
 *     aqe_grouped_finish(ctx, q, kmin, nbins, dev_bins, stream, out, cap, &n_groups)   synchronises `stream`
 * aqe_reduce_grouped is exactly this with a world of one. */
AQE_API int aqe_group_key_range(aqe_ctx* ctx, int group_column, int32_t* key_min, int32_t* key_max);
AQE_API int aqe_grouped_enqueue_bins(aqe_ctx* ctx, const aqe_query* q, int group_column, int32_t key_min, uint32_t nbins, double* dev_bins,
                                     void* stream);
AQE_API int aqe_grouped_finish(aqe_ctx* ctx, const aqe_query* q, int32_t key_min, uint32_t nbins, const double* dev_bins, void* stream,
                               aqe_group_result* out, uint32_t cap, uint32_t* n_groups);

/* ---- stepwise / multi-GPU form ----------------------------------------------------------------
 * One process per GPU; each rank plans the same query over its own shard.  Per round:
 *     aqe_plan_enqueue_round(plan, r, dev_vec, stream)     this shard's partial moment vector
 *     <all-reduce SUM of AQE_MOMENT_VEC doubles, e.g. torch.distributed over RCCL>
 *     aqe_plan_enqueue_update(plan, r, dev_vec, stream)    fold + CLT rules + should_stop (on device)
 * then aqe_plan_enqueue_finalize and aqe_plan_fetch.  Every rank sees the same reduced vector, takes
 * the same stop decision, and a round enqueued after the stop is a device-side no-op.
 * `stream` is a hipStream_t passed as void* (NULL = the context's own stream, a non-blocking stream: it is NOT
 * ordered against a framework's default/null stream — pass the explicit stream your collectives run on). */
#define AQE_MOMENT_VEC 8 /* {n_a, S_a-c n_a, Q_a (shifted), n_b, S_b.., Q_b.., visited, 0}: a = group 0, b = group 1; c = aqe_table_info.shift, moved into
                            the WHERE range when the query has one and the table's shift lies outside it (the same on every shard) */
AQE_API int aqe_plan_create(aqe_ctx* ctx, const aqe_query* q, aqe_plan** out);
/* A single-round plan over caller-given families instead of a sampler's own: `q` supplies the aggregate, the estimator
 * convention, sample_percent and the WHERE range (its method is ignored), `global_samples` the rows the families take over
 * the WHOLE table on all ranks (bookkeeping: the estimators use the rows actually folded, all-reduced by the caller).  on_sorted == 0: rows of the table in global numbering, clipped
 * to this context's shard; on_sorted != 0: positions in THIS context's amount-sorted column (local numbering, see
 * aqe_sorted_counts).  Families must be plain (no AQE_F_PAIR / AQE_F_TOPUP), group 0, with rows
 * ascending in the ordinal (pitch > (seg_len - 1) * step when there is more than one segment) and inside the table: anything else is
 * AQE_ERR_INVALID, nothing reaches a kernel. */
AQE_API int aqe_plan_create_families(aqe_ctx* ctx, const aqe_query* q, const aqe_family* fams, uint32_t n_fams,
                                     uint64_t global_samples, int on_sorted, aqe_plan** out);
AQE_API void aqe_plan_destroy(aqe_plan* plan);
AQE_API int aqe_plan_rounds(const aqe_plan* plan, uint32_t* rounds, int32_t* has_topup);
AQE_API int aqe_plan_enqueue_round(aqe_plan* plan, uint32_t round, double* dev_vec, void* stream);
AQE_API int aqe_plan_enqueue_update(aqe_plan* plan, uint32_t round, const double* dev_vec, void* stream);
AQE_API int aqe_plan_enqueue_finalize(aqe_plan* plan, void* stream);
/* Batched multi-GPU form (plans of 2..32 rounds): ONE launch sweeps every round speculatively and writes
 * this shard's total per round (aqe_plan_totals_len doubles: AQE_MOMENT_VEC per round, in order); ONE
 * all-reduce SUM of that vector; aqe_plan_enqueue_replay then replays the stop rules on the reduced totals
 * and writes the result.  One collective per query instead of one per convergence step — the stop decision
 * is a pure function of the reduced per-round totals, so the answer is identical; what is given up is not
 * sweeping the rounds after the stop.  The reference's top-up (DB.cpp:1031-1040: fewer than base/4 rows
 * collected) is NOT swept speculatively: when it is due the fetched result carries topup_pending = 1 and
 * the caller finishes with the stepwise calls for step r = rounds:
 *     aqe_plan_enqueue_round(plan, rounds, vec) -> all-reduce -> aqe_plan_enqueue_update(plan, rounds, vec)
 *     -> aqe_plan_enqueue_finalize -> aqe_plan_fetch
 * (every rank sees the same mark, so every rank takes the same path).
 * totals_len == 0: the plan has no batched form (single-round or > 32 rounds). */
AQE_API int aqe_plan_totals_len(const aqe_plan* plan, uint32_t* n_doubles);
AQE_API int aqe_plan_enqueue_sweep_totals(aqe_plan* plan, double* dev_totals, void* stream);
AQE_API int aqe_plan_enqueue_replay(aqe_plan* plan, const double* dev_totals, void* stream);
/* A batch of plans of ONE context driven through the batched form together, so that one collective serves all
 * of them and the host pays a few calls per step instead of two per query.  The sweeps of the whole batch are ONE
 * launch (a group of workgroups per plan, see aqe_batch_enqueue_all) on a side stream the context owns; the caller's
 * `stream` — where it issues the collective — is made to wait for it, and after the collective one launch on
 * `stream` replays every plan; the side stream waits for that before it sweeps again.
 *     aqe_batch_enqueue_sweeps(b, totals, row_stride)            row i = plan i's round totals
 *     aqe_batch_join(b, stream)                                  `stream` waits for the batch's sweeps
 *     <ONE all-reduce SUM of the whole [n, row_stride] buffer on `stream`>
 *     aqe_batch_enqueue_replays(b, totals, row_stride, stream)   one replay launch on `stream` for all plans
 *     ... next step ...   aqe_batch_fetch(b, results) waits for the replays (their event) and returns every plan's result
 * Two batches (each with its own buffer) can be software-pipelined — sweeps of B, then join/collective/replays of
 * A, then sweeps of A, ... — so that one batch's collective runs under the other's sweeps.
 * (topup_pending results are finished per plan with the stepwise calls, as above). */
typedef struct aqe_batch aqe_batch;
AQE_API int aqe_batch_create(aqe_plan* const* plans, uint32_t n, aqe_batch** out);
AQE_API void aqe_batch_destroy(aqe_batch* batch);
AQE_API int aqe_batch_enqueue_sweeps(aqe_batch* batch, double* dev_totals, uint64_t row_stride_doubles);
AQE_API int aqe_batch_join(aqe_batch* batch, void* stream);
AQE_API int aqe_batch_enqueue_replays(aqe_batch* batch, const double* dev_totals, uint64_t row_stride_doubles, void* stream);
AQE_API int aqe_batch_fetch(aqe_batch* batch, aqe_result* out_n);
/* Single-GPU form of a batch: Q independent queries in ONE launch, decisions taken in the kernel.  The grid is cut
 * into one group of workgroups per plan (sizes in proportion to the plans' rows); group i runs plan i exactly as
 * aqe_plan_enqueue_all would on a launch of its own — the last workgroup of the group to finish judges the query
 * (lean groups, when every plan's families are plain runs of rows), or the group has its own monitor wave and
 * should_stop word — so the start of the launch and the decision tails of all Q queries are paid once instead of Q times.  This is what
 * replaces the reference's thread creation per call (std::async workers per query, DB.cpp:918-1029) when queries
 * arrive in batches.  A plan predicted to stop early takes its head form (first rounds + the top-up) as its group.
 * Any plan with a family sampler and 1..32 rounds qualifies (not RANDOM_POINTER / RANDOM_DEVICE); the context must
 * hold the whole table.  Results: aqe_batch_fetch (each plan's result is picked up as soon as its group has written
 * it).  A plan has ONE state and ONE result block: it may be part of several batches, but only one execution of it —
 * through a batch or on its own — may be in flight at a time (fetch before enqueueing it again elsewhere).
 * aqe_batch_enqueue_sweeps is the same launch with the decisions left to the replay after the all-reduce. */
AQE_API int aqe_batch_enqueue_all(aqe_batch* batch, void* stream);
/* Timing of the one-launch forms for roofline reports: with profiling on, the launch carries an event pair on its
 * dispatch (the kernel's own begin/end timestamps); aqe_batch_launch_info returns the duration of the most recent
 * launch, the rows it sweeps (all plans; 8 B each) and its workgroups.  Any of the outputs may be NULL. */
AQE_API int aqe_batch_set_profiling(aqe_batch* batch, int enable);
AQE_API int aqe_batch_launch_info(aqe_batch* batch, float* ms, uint64_t* samples, uint32_t* workgroups);
/* ---- the collective behind the C ABI: RCCL over xGMI ---------------------------------------------
 * One all-reduce SUM of the moment vectors replaces the reference's in-process merges (mutex-guarded vector, CAS on
 * atomic<double>, DB.cpp:948-951, 966-967, 2031-2036).  librccl is opened on first use (no link-time dependency; a
 * copy already loaded into the process — PyTorch's — is taken when there is one; AQE_RCCL_LIB overrides).
 *   one process per GPU:   rank 0 calls aqe_comm_unique_id and hands the 128 bytes to every rank out of band (file,
 *                          socket, MPI, a torch store); every rank calls aqe_comm_create(ctx, id, nranks, rank).
 *   one process, n GPUs:   aqe_comm_create_all(ctxs, n, comms) — one context per GPU (ncclCommInitAll); collective
 *                          calls of one step are then bracketed by aqe_comm_group_start / aqe_comm_group_end.
 * Collectives are in place on f64 device memory and are enqueued on `stream` (NULL = the context's own stream). */
#define AQE_COMM_ID_BYTES 128
typedef struct aqe_comm aqe_comm;
AQE_API int aqe_comm_unique_id(void* id128);
AQE_API int aqe_comm_create(aqe_ctx* ctx, const void* id128, int nranks, int rank, aqe_comm** out);
AQE_API int aqe_comm_create_all(aqe_ctx* const* ctxs, int n, aqe_comm** out_n);
AQE_API void aqe_comm_destroy(aqe_comm* comm);
AQE_API int aqe_comm_info(const aqe_comm* comm, int* nranks, int* rank);
AQE_API int aqe_comm_all_reduce_sum(aqe_comm* comm, double* dev_buf, uint64_t count, void* stream);
AQE_API int aqe_comm_all_reduce_max(aqe_comm* comm, double* dev_buf, uint64_t count, void* stream);
AQE_API int aqe_comm_group_start(void);
AQE_API int aqe_comm_group_end(void);
/* A whole query over the ranks of a communicator (what distributed.ShardedQuery.run does from Python): the batched
 * form when the plan has one (ONE collective), else one collective per convergence step; a due top-up is finished
 * with the stepwise step.  dev_vec: max(AQE_MOMENT_VEC, totals_len) doubles of device memory.  Synchronous. */
AQE_API int aqe_plan_run_sharded(aqe_plan* plan, aqe_comm* comm, double* dev_vec, void* stream, aqe_result* out);
/* One step of a batch over the ranks of a communicator, asynchronously: sweeps (ONE launch), join, ONE all-reduce SUM
 * of dev_totals[n_plans][row_stride], replays (ONE launch).  Results: aqe_batch_fetch. */
AQE_API int aqe_batch_run_sharded(aqe_batch* batch, aqe_comm* comm, double* dev_totals, uint64_t row_stride_doubles, uint32_t n_plans, void* stream);

/* ---- the one-shot peer-mapped all-reduce ---------------------------------------------------------------
 * For the moment vectors of this path a collective is pure latency.  On up to 16 GPUs that can write each other's memory
 * (xGMI peers) every rank owns a MAILBOX in its own HBM; an all-reduce is ONE single-workgroup launch per rank that stores
 * the rank's vector into its slot of every peer's mailbox, raises a flag there, waits for the peers' flags in its own mailbox
 * and adds the slots up in rank order (so every rank holds the same sum, bit for bit — what the shared stop decision needs).
 * Replaces, like aqe_comm_*: the reference's in-process merges of its workers' results — the mutex-guarded vector, the
 * future.get() concatenation, the CAS loop on atomic<double> (DB.cpp:948-951, 966-967, 2031-2036) — and the atomic<bool>
 * should_stop every worker polls (DB.cpp:930, 987): every rank derives the same decision from the same sum.
 * A drop-in for aqe_comm_all_reduce_sum between the sweep and the fold:
 *     aqe_mailbox_create(ctx, nranks, rank, &mb)
 *     one process per GPU:  aqe_mailbox_handle(mb, h) -> exchange the 64-byte handles out of band, in rank order ->
 *                           aqe_mailbox_connect(mb, all_handles)            (HIP IPC)
 *     one process, n GPUs:  aqe_mailbox_connect_local(mbs, n)               (peer access)
 *     aqe_mailbox_all_reduce_sum(mb, dev_vec, count, stream)                asynchronous, count <= AQE_MAILBOX_MAX_DOUBLES
 * Every rank must issue the same sequence of calls (the call count is the epoch).  A rank that does not show up within
 * ~2 s ends the peers' launches with the vector untouched and the late ranks' bits in aqe_mailbox_status — never a hang.
 * Destroy a mailbox only after every rank is done with it, and before its context. */
typedef struct aqe_mailbox aqe_mailbox;
#define AQE_MAILBOX_HANDLE_BYTES 64
#define AQE_MAILBOX_MAX_DOUBLES 4096
#define AQE_MAILBOX_MAX_RANKS 16
AQE_API int aqe_mailbox_create(aqe_ctx* ctx, int nranks, int rank, aqe_mailbox** out);
AQE_API int aqe_mailbox_handle(aqe_mailbox* mb, void* handle64);
AQE_API int aqe_mailbox_connect(aqe_mailbox* mb, const void* handles_in_rank_order);
AQE_API int aqe_mailbox_connect_local(aqe_mailbox* const* mbs, int n);
AQE_API int aqe_mailbox_all_reduce_sum(aqe_mailbox* mb, double* dev_buf, uint64_t count, void* stream);
AQE_API int aqe_mailbox_info(const aqe_mailbox* mb, int* nranks, int* rank);
AQE_API int aqe_mailbox_status(aqe_mailbox* mb, uint32_t* late_ranks);
/* A communicator whose SUM all-reduces go through a connected mailbox instead of RCCL (vectors of at most
 * AQE_MAILBOX_MAX_DOUBLES doubles): what aqe_plan_run_sharded / aqe_batch_run_sharded take, so a C or C++ host drives whole
 * sharded queries over the peer-mapped path with the same two calls.  Destroy it (aqe_comm_destroy) before the mailbox. */
AQE_API int aqe_comm_create_mailbox(aqe_ctx* ctx, aqe_mailbox* mb, aqe_comm** out);
AQE_API void aqe_mailbox_destroy(aqe_mailbox* mb);

/* fused single-GPU form: the whole query, asynchronously.  A multi-round (CLT) plan is ONE launch with in-kernel
 * decisions; the reference's top-up (DB.cpp:1031-1040), rarely due, gets its own launch only when the plan's
 * previous execution needed it — otherwise aqe_plan_fetch runs it if the result turns out to want it.
 * A query PREDICTED to stop early (from the coefficient of variation of the table's first rows and the error rule)
 * is launched on a few workgroups over its first rounds and the top-up only; if it has not stopped by then,
 * aqe_plan_fetch launches the remaining rounds and the plan uses the full launch from its next execution on.
 * Either way the answer is the one the round-by-round form gives (sums to rounding). */
AQE_API int aqe_plan_enqueue_all(aqe_plan* plan, void* stream);
AQE_API int aqe_plan_reset(aqe_plan* plan, void* stream); /* re-arm a plan for another execution */
/* Waits for the plan's last execution and returns its result.  After aqe_plan_enqueue_all the result is taken from
 * the plan's pinned result block as soon as the finishing launch has written all of it (a check word over every field
 * says when) — a few microseconds before that launch has drained, so `stream` need not be idle on return; any other
 * way of running the plan, and AQE_NO_POLL=1 in the environment, waits for the stream. */
AQE_API int aqe_plan_fetch(aqe_plan* plan, aqe_result* out, void* stream);
/* device time between the first and last kernel of the most recent execution (HIP events) */
AQE_API int aqe_plan_last_kernel_ms(aqe_plan* plan, float* ms);
/* Per-launch timing for roofline reports: with profiling on, every sweep launch (rounds, top-up) of the
 * next executions carries its own HIP event pair on the launch stream, attached to the dispatch so that it
 * reads the kernel's begin/end timestamps (hipExtLaunchKernelGGL); aqe_plan_launch_ms returns the duration
 * of each launch of the most recent execution, in launch order. */
AQE_API int aqe_plan_set_profiling(aqe_plan* plan, int enable);
AQE_API int aqe_plan_launch_ms(aqe_plan* plan, float* ms, uint32_t cap, uint32_t* n_out);
/* Which kernel swept the rounds of the plan's most recent execution (diagnostics, roofline reports). */
#define AQE_KERNEL_ROUND 0         /* one k_round launch per round                                                 */
#define AQE_KERNEL_SWEEP_PERSIST 1 /* k_sweep_persist: every round in one launch, a monitor wave judges as rounds complete */
#define AQE_KERNEL_SWEEP_LEAN 2    /* k_sweep_lean: every round in one launch, judged once by the last workgroup to arrive */
#define AQE_KERNEL_SWEEP_MULTI 3   /* k_sweep_multi: the plan ran as a group of a batch's one launch (groups with monitor waves) */
#define AQE_KERNEL_SWEEP_LEAN_MULTI 4 /* k_sweep_lean_multi: ... as a lean group (every plan of the batch qualifies)    */
#define AQE_KERNEL_INDEXED 5       /* k_indexed: the seeded-random sampler over its host-built index list (one launch)  */
#define AQE_KERNEL_PERMUTED 6      /* k_permuted: AQE_M_RANDOM_DEVICE, rows drawn in the kernel (one launch)            */
AQE_API int aqe_plan_last_kernel(const aqe_plan* plan, int* kernel);
/* samples (sampled rows) each sweep launch of this shard folds, in launch order; the top-up entry is
 * its upper bound */
AQE_API int aqe_plan_launch_samples(const aqe_plan* plan, uint64_t* samples, uint32_t cap, uint32_t* n_out);

#ifdef __cplusplus
}
#endif
#endif /* AQE_HIP_H */
