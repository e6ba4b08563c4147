// tools/exp_l2rate.hip — scratch micro-benchmark (not product): what rate can loads that are SERVED BY THE L2s reach on this part,
// as a function of waves per compute unit and loads in flight per lane?  Models the bench batch: workgroup b reads the whole
// slice b % 8 (4 MB: resident in that die's L2, read by every workgroup of the die), folding (n, S, Q) as the sweep does.
//   hipcc -O3 --offload-arch=gfx950 -o tools/exp_l2rate.bin tools/exp_l2rate.hip && tools/exp_l2rate.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

struct __attribute__((packed, aligned(8))) Row2 { double x, y; };

template <int THREADS, int UNROLL, int MINBLK>
__global__ __launch_bounds__(THREADS, MINBLK) void k_l2(const double* __restrict__ col, size_t slice_rows, double shift, double* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int WAVES = THREADS / 64;
    const double* base = col + static_cast<size_t>(blockIdx.x % 8) * slice_rows;
    const size_t tile_rows = 128 * UNROLL, ntiles = slice_rows / tile_rows;
    double s = 0.0, q = 0.0;
    for (size_t t = wave; t < ntiles; t += WAVES) {
        const Row2* p = reinterpret_cast<const Row2*>(base + t * tile_rows) + lane;
        Row2 v[UNROLL];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) v[k] = p[k * 64];
#pragma unroll
        for (int k = 0; k < UNROLL; ++k) {
            const double dx = v[k].x - shift, dy = v[k].y - shift;
            s += dx; q += dx * dx;
            s += dy; q += dy * dy;
        }
    }
    if (s == 12345.678 && q == 1.0) out[blockIdx.x] = s;  // (keeps the loads alive)
}

template <int THREADS, int UNROLL, int MINBLK>
void run(const char* name, const double* col, size_t slice_rows, double* out, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> ms;
    for (int it = 0; it < 12; ++it) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_l2<THREADS, UNROLL, MINBLK>), dim3(grid), dim3(THREADS), 0, 0, col, slice_rows, 500.5, out);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float m; hipEventElapsedTime(&m, e0, e1);
        if (it >= 2) ms.push_back(m);
    }
    std::sort(ms.begin(), ms.end());
    const double bytes = static_cast<double>(grid) * slice_rows * 8.0;
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, k_l2<THREADS, UNROLL, MINBLK>, THREADS, 0);
    printf("%-34s grid %5d  blocks/CU(max) %d  waves/CU %2d  median %8.2f us  executed %7.1f MB  rate %6.2f TB/s\n", name, grid, nb, nb * THREADS / 64,
           1e3 * ms[ms.size() / 2], bytes / 1e6, bytes / (ms[ms.size() / 2] * 1e-3) / 1e12);
}

int main() {
    const size_t slice_rows = 512 * 1024;  // 4 MB per slice
    double* col; double* out;
    hipMalloc(&col, slice_rows * 8 * 8);
    hipMalloc(&out, 8192 * 8);
    std::vector<double> h(slice_rows * 8);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + (i * 2654435761u % 999);
    hipMemcpy(col, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    // the product's shape: 256 workgroups of 16 waves, 8 loads in flight per lane
    run<1024, 8, 1>("1024 thr x1/CU, 8 loads", col, slice_rows, out, 256);
    run<1024, 4, 1>("1024 thr x1/CU, 4 loads", col, slice_rows, out, 256);
    run<1024, 16, 1>("1024 thr x1/CU, 16 loads", col, slice_rows, out, 256);
    // more waves per compute unit (each workgroup still reads a whole slice: executed bytes grow with the grid)
    run<512, 8, 3>("512 thr x3/CU, 8 loads", col, slice_rows, out, 768);
    run<512, 8, 4>("512 thr x4/CU, 8 loads", col, slice_rows, out, 1024);
    run<256, 8, 5>("256 thr x5/CU, 8 loads", col, slice_rows, out, 1280);
    run<256, 8, 6>("256 thr x6/CU, 8 loads", col, slice_rows, out, 1536);
    run<256, 8, 8>("256 thr x8/CU, 8 loads", col, slice_rows, out, 2048);
    run<256, 4, 8>("256 thr x8/CU, 4 loads", col, slice_rows, out, 2048);
    run<256, 16, 4>("256 thr x4/CU, 16 loads", col, slice_rows, out, 1024);
    run<256, 16, 8>("256 thr x8/CU, 16 loads", col, slice_rows, out, 2048);
    return 0;
}
