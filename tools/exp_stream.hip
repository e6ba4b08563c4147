// tools/exp_stream.hip — scratch micro-benchmark (not product): a single-round sweep of X MB (every tile read ONCE, dealt out wave by
// wave across the launch), by workgroup shape / workgroups per compute unit / loads in flight: is a cache-resident 80 / 160 MB
// sweep (Infinity Cache) or an 8 GB one (HBM) faster with more waves per compute unit than the product's 16?
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp_stream.bin tools/exp_stream.hip && tools/exp_stream.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct __attribute__((packed, aligned(8))) Row2 { double x, y; };

template <int THREADS, int UNROLL, int MINBLK, bool NT>
__global__ __launch_bounds__(THREADS, MINBLK) void k_stream(const double* __restrict__ col, size_t rows, double shift, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVES = THREADS / 64;
    const size_t tile_rows = 128 * UNROLL, ntiles = rows / tile_rows;
    double s = 0.0, q = 0.0;
    for (size_t t = static_cast<size_t>(blockIdx.x) * WAVES + wave; t < ntiles; t += static_cast<size_t>(gridDim.x) * WAVES) {
        const Row2* p = reinterpret_cast<const Row2*>(col + t * tile_rows) + lane;
        Row2 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            if (NT) { v[k].x = __builtin_nontemporal_load(&p[k * 64].x); v[k].y = __builtin_nontemporal_load(&p[k * 64].y); }
            else v[k] = p[k * 64];
        }
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const double dx = v[k].x - shift, dy = v[k].y - shift;
            s += dx; q += dx * dx;
            s += dy; q += dy * dy;
        }
    }
    if (s == 12345.678 && q == 1.0) out[blockIdx.x] = s;
}

template <int THREADS, int UNROLL, int MINBLK, bool NT>
void run(const char* name, const double* col, size_t rows, double* out, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_stream<THREADS, UNROLL, MINBLK, NT>), dim3(grid), dim3(THREADS), 0, 0, col, rows, 500.5, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float m; hipEventElapsedTime(&m, e0, e1);
        if (it >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = rows * 8.0;
    printf("  %-30s grid %5d  median %9.2f us  %7.2f TB/s  (%.3f of 8)\n", name, grid, 1e3 * ms[ms.size() / 2], bytes / (ms[ms.size() / 2] * 1e-3) / 1e12,
           bytes / (ms[ms.size() / 2] * 1e-3) / 8e12);
}

int main() {
    const size_t max_rows = 1000000000ull;
    double* col; double* out;
    if (hipMalloc(&col, max_rows * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMalloc(&out, 8192 * 8);
    hipMemset(col, 0, max_rows * 8);
    hipDeviceSynchronize();
    for (size_t rows : {size_t(10000000), size_t(20000000), size_t(100000000), size_t(1000000000)}) {
        printf("%zu rows = %.0f MB\n", rows, rows * 8.0 / 1e6);
        const bool big = rows * 8 > (256u << 20);
        if (!big) {
            run<1024, 8, 1, false>("1024 thr x1/CU, 8 loads", col, rows, out, 256);
            run<512, 8, 3, false>("512 thr x3/CU, 8 loads", col, rows, out, 768);
            run<256, 8, 4, false>("256 thr x4/CU, 8 loads", col, rows, out, 1024);
            run<256, 8, 6, false>("256 thr x6/CU, 8 loads", col, rows, out, 1536);
            run<256, 8, 8, false>("256 thr x8/CU, 8 loads", col, rows, out, 2048);
            run<256, 4, 8, false>("256 thr x8/CU, 4 loads", col, rows, out, 2048);
            run<256, 16, 4, false>("256 thr x4/CU, 16 loads", col, rows, out, 1024);
        } else {
            run<1024, 8, 1, true>("1024 thr x1/CU, 8 loads nt", col, rows, out, 256);
            run<512, 8, 3, true>("512 thr x3/CU, 8 loads nt", col, rows, out, 768);
            run<256, 8, 6, true>("256 thr x6/CU, 8 loads nt", col, rows, out, 1536);
            run<256, 8, 8, true>("256 thr x8/CU, 8 loads nt", col, rows, out, 2048);
            run<256, 4, 8, true>("256 thr x8/CU, 4 loads nt", col, rows, out, 2048);
            run<256, 16, 4, true>("256 thr x4/CU, 16 loads nt", col, rows, out, 1024);
        }
    }
    return 0;
}
