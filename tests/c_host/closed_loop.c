/* tests/c_host/closed_loop.c — closed-loop latency of one query through the C ABI from a plain-C host (no Python, no
 * ctypes, no torch): enqueue -> fetch, one query in flight, wall clock around the pair.  What a C/C++ maintainer of the
 * reference's bindings (bindings.cpp:10-137) would see per call.  bench.py builds and runs it beside its Python loop.
 *     closed_loop <rows> <reps>   ->  one JSON object per query shape on stdout */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "aqe_hip.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc__ = (call);                                                                            \
        if (rc__ != AQE_OK) {                                                                         \
            fprintf(stderr, "%s -> %d (%s): %s\n", #call, rc__, aqe_status_string(rc__), aqe_last_error(ctx)); \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static double now_us(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return 1e6 * (double)ts.tv_sec + 1e-3 * (double)ts.tv_nsec;
}

static int cmp(const void* a, const void* b) { return (*(const double*)a > *(const double*)b) - (*(const double*)a < *(const double*)b); }

int main(int argc, char** argv) {
    const uint64_t rows = argc > 1 ? strtoull(argv[1], NULL, 10) : 10000000ull;
    const int reps = argc > 2 ? atoi(argv[2]) : 200;
    aqe_ctx* ctx = NULL;
    if (aqe_abi_version() != AQE_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 1; }
    CHECK(aqe_create(0, &ctx));
    CHECK(aqe_generate_synthetic(ctx, rows, 0, rows, 42, 0));
    const char* names[3] = {"CLT AVG e=0.01% (the bench query)", "stride 1% SUM", "exact SUM"};
    double* lat = (double*)malloc(sizeof(double) * (size_t)reps);
    if (!lat) return 1;
    for (int k = 0; k < 3; ++k) {
        aqe_query q;
        aqe_query_defaults(&q);
        if (k == 0) {
            q.method = AQE_M_CLT_DUAL_POINTER; q.agg = AQE_AVG; q.sample_percent = aqe_error_to_sample_percent(0.01);
            q.max_error_percent = 0.01; q.num_threads = 4; q.clt_round0 = 4096; q.clt_growth = 4;
        } else if (k == 1) {
            q.method = AQE_M_MEMORY_STRIDE; q.sample_percent = 1.0;
        } else {
            q.method = AQE_M_EXACT; q.sample_percent = 100.0;
        }
        aqe_plan* plan = NULL;
        aqe_result r;
        CHECK(aqe_plan_create(ctx, &q, &plan));
        for (int i = 0; i < 10; ++i) { CHECK(aqe_plan_enqueue_all(plan, NULL)); CHECK(aqe_plan_fetch(plan, &r, NULL)); }
        for (int i = 0; i < reps; ++i) {
            const double t0 = now_us();
            CHECK(aqe_plan_enqueue_all(plan, NULL));
            CHECK(aqe_plan_fetch(plan, &r, NULL));
            lat[i] = now_us() - t0;
        }
        qsort(lat, (size_t)reps, sizeof(double), cmp);
        printf("{\"query\": \"%s\", \"rows\": %llu, \"reps\": %d, \"p50_us\": %.3f, \"min_us\": %.3f, \"p90_us\": %.3f, \"n\": %llu, \"value\": %.17g}\n",
               names[k], (unsigned long long)rows, reps, lat[reps / 2], lat[0], lat[(reps * 9) / 10], (unsigned long long)r.n, r.value);
        aqe_plan_destroy(plan);
    }
    free(lat);
    aqe_destroy(ctx);
    return 0;
}
