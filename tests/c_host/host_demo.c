/* tests/c_host/host_demo.c — a plain-C host of include/aqe_hip.h (no HIP headers, no Python, no torch): what
 * INTEGRATION.md §C describes.  Generates a table, runs a CLT query three ways — aqe_reduce, a batch of queries in ONE
 * launch, and the sharded path through the library's own RCCL communicator (a world of one rank) — and checks that
 * they agree.  Built and run by tests/test_gpu_parity.py::test_plain_c_host_program (gcc, links libaqe_hip.so only). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "aqe_hip.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc__ = (call);                                                                            \
        if (rc__ != AQE_OK) {                                                                         \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc__, aqe_status_string(rc__), aqe_last_error(ctx)); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static int same(const aqe_result* a, const aqe_result* b) {
    return a->n == b->n && a->visited == b->visited && a->converged == b->converged && a->rounds == b->rounds && a->topup == b->topup &&
           fabs(a->sum - b->sum) <= 1e-12 * fabs(b->sum) && fabs(a->value - b->value) <= 1e-12 * fabs(b->value) &&
           fabs(a->ci_lower - b->ci_lower) <= 1e-12 * fabs(b->ci_lower);
}

int main(int argc, char** argv) {
    const uint64_t rows = argc > 1 ? strtoull(argv[1], NULL, 10) : 1000000ull;
    aqe_ctx* ctx = NULL;
    if (aqe_abi_version() != AQE_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    CHECK(aqe_create(0, &ctx));
    CHECK(aqe_generate_synthetic(ctx, rows, 0, rows, 42, 0));

    enum { Q = 6 };
    aqe_query q[Q];
    aqe_result want[Q], got[Q];
    for (int i = 0; i < Q; ++i) {
        aqe_query_defaults(&q[i]);
        if (i < 4) {                                    /* CLT monitor: never converging, early stop + top-up, in between */
            const double e[4] = {0.0, 1.0, 0.3, 0.01};
            q[i].method = AQE_M_CLT_DUAL_POINTER;
            q[i].agg = AQE_AVG;
            q[i].sample_percent = aqe_error_to_sample_percent(e[i] > 0 ? e[i] : 0.01);
            q[i].max_error_percent = e[i];
            q[i].num_threads = 4 + 2 * i;
            q[i].clt_round0 = 1024;
            q[i].clt_growth = 4;
        } else if (i == 4) {
            q[i].method = AQE_M_MEMORY_STRIDE; q[i].sample_percent = 20.0;
        } else {
            q[i].method = AQE_M_BLOCK; q[i].sample_percent = 5.0; q[i].has_where = 1; q[i].where_min = 250.0; q[i].where_max = 750.0;
            q[i].convention = AQE_EST_CPP;
        }
        CHECK(aqe_reduce(ctx, &q[i], &want[i]));
    }

    /* a batch of different queries in ONE launch */
    aqe_plan* plans[Q];
    for (int i = 0; i < Q; ++i) CHECK(aqe_plan_create(ctx, &q[i], &plans[i]));
    aqe_batch* batch = NULL;
    CHECK(aqe_batch_create(plans, Q, &batch));
    for (int step = 0; step < 3; ++step) {
        CHECK(aqe_batch_enqueue_all(batch, NULL));
        CHECK(aqe_batch_fetch(batch, got));
        for (int i = 0; i < Q; ++i)
            if (!same(&got[i], &want[i])) { fprintf(stderr, "batch: query %d differs (step %d)\n", i, step); return 1; }
    }

    /* the sharded path through the library's own RCCL communicator (here: one rank) */
    char id[AQE_COMM_ID_BYTES];
    aqe_comm* comm = NULL;
    double* vec = NULL;
    int nranks = 0, rank = -1;
    CHECK(aqe_comm_unique_id(id));
    CHECK(aqe_comm_create(ctx, id, 1, 0, &comm));
    CHECK(aqe_comm_info(comm, &nranks, &rank));
    CHECK(aqe_device_malloc(ctx, sizeof(double) * 8 * 64, (void**)&vec));
    for (int i = 0; i < Q; ++i) {
        aqe_result r;
        CHECK(aqe_plan_run_sharded(plans[i], comm, vec, NULL, &r));
        if (!same(&r, &want[i]) || r.topup_pending) { fprintf(stderr, "sharded: query %d differs\n", i); return 1; }
    }
    double* totals = NULL;
    CHECK(aqe_device_malloc(ctx, sizeof(double) * 4 * 8 * 64, (void**)&totals));
    aqe_batch* b4 = NULL;
    CHECK(aqe_batch_create(plans, 4, &b4));  /* the four CLT plans: a batch per collective */
    CHECK(aqe_batch_run_sharded(b4, comm, totals, 8 * 64, 4, NULL));
    CHECK(aqe_batch_fetch(b4, got));
    for (int i = 0; i < 4; ++i)
        if (got[i].converged != want[i].converged || got[i].rounds != want[i].rounds || (got[i].topup_pending != 0) != (want[i].topup != 0)) {
            fprintf(stderr, "sharded batch: query %d differs\n", i);
            return 1;
        }
    /* what the ranks of a sharded table exchange for the variance-aware samplers, from C (here the "shard" is the whole table):
     * zone moments -> variances -> adaptive plan; a value's rank in the sorted column; a query over caller-given families */
    {
        double zm[30], var[10];
        CHECK(aqe_zone_moments(ctx, zm));
        for (int z = 0; z < 10; ++z) { const double m = zm[3 * z + 1] / zm[3 * z]; var[z] = zm[3 * z + 2] / zm[3 * z] - m * m; }
        CHECK(aqe_set_zone_variances(ctx, var));
        aqe_query qa; aqe_query_defaults(&qa); qa.method = AQE_M_ADAPTIVE_BLOCK; qa.sample_percent = 2.0; qa.block_size = 500;
        aqe_result ra; CHECK(aqe_reduce(ctx, &qa, &ra));
        const double probe[3] = {-1.0, 500.5, 1e9};
        uint64_t lt[3], le[3];
        CHECK(aqe_sorted_counts(ctx, probe, 3, lt, le));
        if (ra.n == 0 || lt[0] != 0 || le[2] != rows || lt[1] > le[1] || le[1] == 0 || le[1] >= rows) { fprintf(stderr, "variance-aware pieces differ\n"); return 1; }
        aqe_family f; memset(&f, 0, sizeof f);
        f.row0 = le[1]; f.seg_len = 1000; f.step = 1; f.ord_hi = 1000;   /* the 1000 rows next above 500.5 in the sorted column */
        aqe_query qe; aqe_query_defaults(&qe); qe.method = AQE_M_EXACT; qe.agg = AQE_AVG;
        aqe_plan* pf = NULL; aqe_result rf;
        if (le[1] + 1000 <= rows) {
            CHECK(aqe_plan_create_families(ctx, &qe, &f, 1, 1000, 1, &pf));
            CHECK(aqe_plan_enqueue_all(pf, NULL));
            CHECK(aqe_plan_fetch(pf, &rf, NULL));
            aqe_plan_destroy(pf);
            if (rf.n != 1000 || !(rf.mean > 500.5) || !(rf.mean < 502.0)) { fprintf(stderr, "families plan: n %llu mean %f\n", (unsigned long long)rf.n, rf.mean); return 1; }
        }
        f.row0 = rows - 10;  /* leaves the table: refused on the host */
        if (aqe_plan_create_families(ctx, &qe, &f, 1, 1000, 0, &pf) != AQE_ERR_INVALID) { fprintf(stderr, "bad family accepted\n"); return 1; }
    }
    /* the peer-mapped mailbox from C (one rank: its own mailbox is the only peer): the vector comes back as it went in */
    {
        aqe_mailbox* mb = NULL;
        char handle[AQE_MAILBOX_HANDLE_BYTES];
        double host_in[8] = {1.5, -2.0, 3.25, 0.0, 1e300, -1e-300, 7.0, 8.0}, host_out[8];
        uint32_t late = 1;
        CHECK(aqe_mailbox_create(ctx, 1, 0, &mb));
        CHECK(aqe_mailbox_handle(mb, handle));   /* what a rank would hand to its peers */
        CHECK(aqe_mailbox_connect(mb, handle));
        for (int rep = 0; rep < 3; ++rep) {      /* three epochs: both parities */
            CHECK(aqe_device_write(ctx, vec, host_in, sizeof host_in, NULL));
            CHECK(aqe_mailbox_all_reduce_sum(mb, vec, 8, NULL));
            CHECK(aqe_device_read(ctx, host_out, vec, sizeof host_out, NULL));
            if (memcmp(host_in, host_out, sizeof host_in) != 0) { fprintf(stderr, "mailbox: vector changed (epoch %d)\n", rep); return 1; }
        }
        CHECK(aqe_mailbox_status(mb, &late));
        if (late != 0) { fprintf(stderr, "mailbox: late ranks %u\n", late); return 1; }
        /* ... and as the collective of whole sharded queries: the same two calls as with the RCCL communicator */
        aqe_comm* mc = NULL;
        CHECK(aqe_comm_create_mailbox(ctx, mb, &mc));
        for (int i = 0; i < Q; ++i) {
            aqe_result r;
            CHECK(aqe_plan_run_sharded(plans[i], mc, vec, NULL, &r));
            if (!same(&r, &want[i]) || r.topup_pending) { fprintf(stderr, "sharded over the mailbox: query %d differs\n", i); return 1; }
        }
        aqe_comm_destroy(mc);
        aqe_mailbox_destroy(mb);
    }
    aqe_batch_destroy(b4);
    aqe_batch_destroy(batch);
    for (int i = 0; i < Q; ++i) aqe_plan_destroy(plans[i]);
    CHECK(aqe_device_free(ctx, totals));
    CHECK(aqe_device_free(ctx, vec));
    aqe_comm_destroy(comm);
    printf("host_demo ok: %d queries, ranks %d, avg %.6f [%.6f, %.6f] n %llu\n", Q, nranks, want[0].value, want[0].ci_lower, want[0].ci_upper,
           (unsigned long long)want[0].n);
    aqe_destroy(ctx);
    return 0;
}
