// tools/exp_access.hip — scratch micro-benchmark (not part of the product): how fast can gfx950 fold
// 40 %-dense samples (rows 0,2 mod 5) out of an 80 MB f64 column, by access form?
//   hipcc -O3 --offload-arch=gfx950 tools/exp_access.hip -o /tmp/exp_access && /tmp/exp_access
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ double wsum(double v) { for (int o = 32; o; o >>= 1) v += __shfl_down(v, o, 64); return v; }
__device__ void finish(double s, double q, double* out) {  // one plain 16-byte partial per wave, no atomics
    s = wsum(s); q = wsum(q);
    if ((threadIdx.x & 63) == 0) { double* p = out + 2 * ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)); p[0] = s; p[1] = q; }
}

// (a) dense: every row, 16-byte loads, U loads in flight per lane
template <int U> __global__ __launch_bounds__(256) void k_dense(const double2* __restrict__ x, u64 n2, double* out) {
    double s = 0, q = 0;
    u64 stride = (u64)gridDim.x * 256 * U;
    for (u64 i0 = (u64)blockIdx.x * 256 * U + threadIdx.x; i0 < n2; i0 += stride) {
        double2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { u64 i = i0 + (u64)k * 256; v[k] = x[i < n2 ? i : 0]; if (i >= n2) v[k] = {0, 0}; }
#pragma unroll
        for (int k = 0; k < U; ++k) { s += v[k].x + v[k].y; q += v[k].x * v[k].x + v[k].y * v[k].y; }
    }
    finish(s, q, out);
}
// (b) gather: one pointer, rows start + k*step, 8-byte loads
template <int U> __global__ __launch_bounds__(256) void k_gather(const double* __restrict__ x, u64 start, u64 step, u64 cnt, double* out) {
    double s = 0, q = 0;
    u64 stride = (u64)gridDim.x * 256 * U;
    for (u64 k0 = (u64)blockIdx.x * 256 * U + threadIdx.x; k0 < cnt; k0 += stride) {
        double v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { u64 o = k0 + (u64)k * 256; v[k] = x[o < cnt ? start + o * step : 0]; if (o >= cnt) v[k] = 0; }
#pragma unroll
        for (int k = 0; k < U; ++k) { s += v[k]; q += v[k] * v[k]; }
    }
    finish(s, q, out);
}
// (c) paired gather: two pointers with the same step in one pass
template <int U> __global__ __launch_bounds__(256) void k_pair(const double* __restrict__ x, u64 s0, u64 s1, u64 step, u64 cnt, double* out) {
    double s = 0, q = 0;
    u64 stride = (u64)gridDim.x * 256 * U;
    for (u64 k0 = (u64)blockIdx.x * 256 * U + threadIdx.x; k0 < cnt; k0 += stride) {
        double v[U], w[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { u64 o = k0 + (u64)k * 256; bool ok = o < cnt; v[k] = x[ok ? s0 + o * step : 0]; w[k] = x[ok ? s1 + o * step : 0]; if (!ok) { v[k] = 0; w[k] = 0; } }
#pragma unroll
        for (int k = 0; k < U; ++k) { s += v[k] + w[k]; q += v[k] * v[k] + w[k] * w[k]; }
    }
    finish(s, q, out);
}
// (d) dense + residue select: read everything coalesced, keep rows with (row % step) in {r0, r1}
template <int U> __global__ __launch_bounds__(256) void k_select(const double2* __restrict__ x, u64 n2, unsigned step, unsigned r0, unsigned r1, double* out) {
    double s = 0, q = 0;
    u64 stride = (u64)gridDim.x * 256 * U;
    for (u64 i0 = (u64)blockIdx.x * 256 * U + threadIdx.x; i0 < n2; i0 += stride) {
        double2 v[U];
#pragma unroll
        for (int k = 0; k < U; ++k) { u64 i = i0 + (u64)k * 256; v[k] = x[i < n2 ? i : 0]; if (i >= n2) v[k] = {0, 0}; }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            u64 row = 2 * (i0 + (u64)k * 256);
            unsigned m0 = (unsigned)(row % step), m1 = (m0 + 1 == step) ? 0 : m0 + 1;
            double a = (m0 == r0 || m0 == r1) ? v[k].x : 0.0, b = (m1 == r0 || m1 == r1) ? v[k].y : 0.0;
            s += a + b; q += a * a + b * b;
        }
    }
    finish(s, q, out);
}

int main() {
    const u64 N = 10000000;
    double* x; double* out;
    CK(hipMalloc(&x, N * 8)); CK(hipMalloc(&out, 16 * 4 * 8192));
    std::vector<double> h(N); for (u64 i = 0; i < N; ++i) h[i] = 1.0 + (double)(i % 997);
    CK(hipMemcpy(x, h.data(), N * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, auto launch, double bytes_alg, double bytes_touched) {
        for (int i = 0; i < 5; ++i) launch();
        (void)hipDeviceSynchronize();
        const int R = 200;
        (void)hipEventRecord(e0);
        for (int i = 0; i < R; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double us = 1e3 * ms / R;
        printf("%-34s %8.2f us/launch  alg %7.1f GB/s  touched %7.1f GB/s\n", name, us, bytes_alg / us / 1e3, bytes_touched / us / 1e3);
    };
    const u64 cnt = N / 5;  // samples per pointer at step 5
    for (int grid : {256, 512, 1024, 2048, 4096}) {
        printf("-- grid %d\n", grid);
        timeit("dense U4 (all rows)", [&] { hipLaunchKernelGGL(k_dense<4>, dim3(grid), dim3(256), 0, 0, (const double2*)x, N / 2, out); }, N * 8.0, N * 8.0);
        timeit("dense U8 (all rows)", [&] { hipLaunchKernelGGL(k_dense<8>, dim3(grid), dim3(256), 0, 0, (const double2*)x, N / 2, out); }, N * 8.0, N * 8.0);
        timeit("gather step5 U8 (1 pointer)", [&] { hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, x, 0ull, 5ull, cnt, out); }, cnt * 8.0, N * 8.0);
        timeit("2x gather step5 U8 (2 launches)", [&] { hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, x, 0ull, 5ull, cnt, out); hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, x, 2ull, 5ull, cnt, out); }, 2 * cnt * 8.0, 2 * N * 8.0);
        timeit("pair gather step5 U4", [&] { hipLaunchKernelGGL(k_pair<4>, dim3(grid), dim3(256), 0, 0, x, 0ull, 2ull, 5ull, cnt, out); }, 2 * cnt * 8.0, N * 8.0);
        timeit("pair gather step5 U8", [&] { hipLaunchKernelGGL(k_pair<8>, dim3(grid), dim3(256), 0, 0, x, 0ull, 2ull, 5ull, cnt, out); }, 2 * cnt * 8.0, N * 8.0);
        timeit("dense select {0,2} mod 5 U4", [&] { hipLaunchKernelGGL(k_select<4>, dim3(grid), dim3(256), 0, 0, (const double2*)x, N / 2, 5u, 0u, 2u, out); }, 2 * cnt * 8.0, N * 8.0);
        timeit("dense select {0,2} mod 5 U8", [&] { hipLaunchKernelGGL(k_select<8>, dim3(grid), dim3(256), 0, 0, (const double2*)x, N / 2, 5u, 0u, 2u, out); }, 2 * cnt * 8.0, N * 8.0);
        timeit("gather step100 U8 (1% stride)", [&] { hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(256), 0, 0, x, 0ull, 100ull, N / 100, out); }, N / 100 * 8.0, N / 100 * 64.0);
    }
    timeit("empty-ish kernel (grid 1)", [&] { hipLaunchKernelGGL(k_gather<8>, dim3(1), dim3(256), 0, 0, x, 0ull, 5ull, 64ull, out); }, 512, 512);
    return 0;
}
