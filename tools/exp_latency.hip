// tools/exp_latency.hip — scratch micro-benchmark (not part of the product): what does ONE launch that
// sweeps X MB of an f64 column and hands seven sums to the host cost from dispatch begin to dispatch end,
// by hand-off protocol?  Launches are separated by a stream synchronise (no overlap between launches), timed
// by the dispatch's own begin/end events (hipExtLaunchKernelGGL) and by s_memrealtime stamps (100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/exp_latency.hip -o tools/exp_latency.bin && tools/exp_latency.bin
//
// hand-off modes (all-sc1: agent-scope stores/loads/atomics, s_waitcnt between data and what publishes it; no fences):
//   0  none: per-workgroup partial stored, nobody folds
//   1  sharded tickets (8 shards + top), the last arriver folds and writes the host result   [k_round's protocol]
//   2  one ticket counter
//   3  flags: partial, drain, flag word; a MONITOR wave (wave 0 of workgroup 0, sweeps nothing) polls the flags,
//      then reads the partials, folds, writes the host result                                [k_sweep_persist's protocol]
//   4  tagged granules: the partial is four 16-byte granules {f64, u32, tag}; one store instruction, no drain, no
//      flag; the monitor polls the granules themselves: the poll that finds every tag current already holds the data
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned long long u64;
typedef double d2 __attribute__((ext_vector_type(2)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
#define RLX __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

__device__ __forceinline__ u64 now() { return __builtin_amdgcn_s_memrealtime(); }
__device__ __forceinline__ double wsum(double v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

struct Args {
    const d2* x;
    u64 tiles;                        // tiles of 512 pairs (8 KiB)
    double* partial;                  // per workgroup: 8 doubles (7 sums + flag word)
    unsigned* tickets;                // [shard * 32], top at [16 * 32]
    double* result;                   // pinned host: 7 sums + epoch
    u64* stamps;                      // per wave: start, loads done, end;  monitor: [3 * V + 0..3] seen, folded, stored
    unsigned epoch;
    int mode;
    int sweepers_only;                // modes 3/4: wave 0 of workgroup 0 is the monitor
};

__device__ __forceinline__ void store_granule(void* p, double v, unsigned w, unsigned tag) {
    u4 g; g.x = (unsigned)__double2loint(v); g.y = (unsigned)__double2hiint(v); g.z = w; g.w = tag;
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(g) : "memory");
}

template <int WAVES> __global__ __launch_bounds__(WAVES * 64) void k_sweep(Args a) {
    __shared__ double lds[WAVES][8];
    __shared__ unsigned lds_cnt;
    if (threadIdx.x == 0) lds_cnt = 0;
    __syncthreads();   // every wave, the monitor too, exactly once
    const u64 t0 = now();
    const unsigned lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool has_monitor = a.mode >= 3;
    const u64 V = (u64)gridDim.x * WAVES - (has_monitor ? 1 : 0);
    const u64 wid = (u64)blockIdx.x * WAVES + w;
    if (has_monitor && wid == 0) {
        // ---- the monitor ----
        __builtin_amdgcn_s_setprio(3);
        const unsigned G = gridDim.x;
        double tot[7] = {0, 0, 0, 0, 0, 0, 0};
        u64 t_seen = 0, t_folded = 0;
        if (a.mode == 3) {
            const u64* fl = reinterpret_cast<const u64*>(a.partial) + 7;
            for (unsigned polls = 0; polls < 2000000u; ++polls) {
                bool ok = true;
                u64 f[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const unsigned g = lane + 64 * k; f[k] = g < G ? __hip_atomic_load(fl + (size_t)g * 8, RLX) : (u64)a.epoch + 1; }
#pragma unroll
                for (int k = 0; k < 4; ++k) ok = ok && f[k] == (u64)a.epoch + 1;
                if (__ballot(!ok) == 0) break;
            }
            t_seen = now();
            // lane 8 j + c: component c of slots j, j + 8, ...
            const unsigned c = lane & 7, j = lane >> 3;
            double run = 0;
            double x[32];
#pragma unroll
            for (int m = 0; m < 32; ++m) { const unsigned g = j + 8 * m; x[m] = g < G ? __hip_atomic_load(a.partial + (size_t)g * 8 + c, RLX) : 0.0; }
#pragma unroll
            for (int m = 0; m < 32; ++m) run += x[m];
            if (c == 7) run = 0;
            // over the 8 slot classes (lane bits 3..5)
            run += __shfl_xor(run, 8, 64); run += __shfl_xor(run, 16, 64); run += __shfl_xor(run, 32, 64);
            t_folded = now();
#pragma unroll
            for (int cc = 0; cc < 7; ++cc) tot[cc] = __shfl(run, cc, 64);
        } else {
            // granule (slot g, part p) at partial + g*64 B + p*16 B; lane L: part L & 3 of slots (L >> 2) + 16 m
            const char* base = reinterpret_cast<const char*>(a.partial);
            u4 g[16];
            const unsigned part = lane & 3, s0 = lane >> 2;
            for (unsigned polls = 0; polls < 2000000u; ++polls) {
                const char* p = base + (size_t)s0 * 64 + part * 16;
                asm volatile(
                    "global_load_dwordx4 %0, %16, off sc1\n global_load_dwordx4 %1, %16, off offset:1024 sc1\n"
                    "global_load_dwordx4 %2, %16, off offset:2048 sc1\n global_load_dwordx4 %3, %16, off offset:3072 sc1\n"
                    "global_load_dwordx4 %4, %17, off sc1\n global_load_dwordx4 %5, %17, off offset:1024 sc1\n"
                    "global_load_dwordx4 %6, %17, off offset:2048 sc1\n global_load_dwordx4 %7, %17, off offset:3072 sc1\n"
                    "global_load_dwordx4 %8, %18, off sc1\n global_load_dwordx4 %9, %18, off offset:1024 sc1\n"
                    "global_load_dwordx4 %10, %18, off offset:2048 sc1\n global_load_dwordx4 %11, %18, off offset:3072 sc1\n"
                    "global_load_dwordx4 %12, %19, off sc1\n global_load_dwordx4 %13, %19, off offset:1024 sc1\n"
                    "global_load_dwordx4 %14, %19, off offset:2048 sc1\n global_load_dwordx4 %15, %19, off offset:3072 sc1\n"
                    "s_waitcnt vmcnt(0)"
                    : "=&v"(g[0]), "=&v"(g[1]), "=&v"(g[2]), "=&v"(g[3]), "=&v"(g[4]), "=&v"(g[5]), "=&v"(g[6]), "=&v"(g[7]),
                      "=&v"(g[8]), "=&v"(g[9]), "=&v"(g[10]), "=&v"(g[11]), "=&v"(g[12]), "=&v"(g[13]), "=&v"(g[14]), "=&v"(g[15])
                    : "v"(p), "v"(p + 4096), "v"(p + 8192), "v"(p + 12288)
                    : "memory");
                bool ok = true;
#pragma unroll
                for (int m = 0; m < 16; ++m) ok = ok && (s0 + 16 * m >= G || g[m].w == a.epoch + 1);
                if (__ballot(!ok) == 0) break;
            }
            t_seen = now();
            double run = 0; unsigned cnt = 0;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const bool in = s0 + 16 * m < G;
                run += in ? __hiloint2double((int)g[m].y, (int)g[m].x) : 0.0;
                cnt += in ? g[m].z : 0u;
            }
            double cn = (double)cnt;
            run += __shfl_xor(run, 4, 64); run += __shfl_xor(run, 8, 64); run += __shfl_xor(run, 16, 64); run += __shfl_xor(run, 32, 64);
            cn += __shfl_xor(cn, 4, 64); cn += __shfl_xor(cn, 8, 64); cn += __shfl_xor(cn, 16, 64); cn += __shfl_xor(cn, 32, 64);
            t_folded = now();
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) tot[cc] = __shfl(run, cc, 64);
#pragma unroll
            for (int cc = 0; cc < 3; ++cc) tot[4 + cc] = __shfl(cn, cc, 64);
        }
        if (lane < 7) {
            double mine = 0;
#pragma unroll
            for (int cc = 0; cc < 7; ++cc) mine = lane == cc ? tot[cc] : mine;
            __hip_atomic_store(a.result + lane, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        if (lane == 0) {
            __hip_atomic_store(a.result + 7, (double)a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            u64* st = a.stamps + 3 * ((u64)gridDim.x * WAVES);
            st[0] = t_seen; st[1] = t_folded; st[2] = now();
            u64* me = a.stamps; me[0] = t0; me[1] = t0; me[2] = t0;  // (skipped by the host)
        }
        return;
    }
    const u64 v = wid - (has_monitor ? 1 : 0);
    double s = 0, q = 0;
    for (u64 t = v; t < a.tiles; t += V) {
        const d2* p = a.x + t * 512 + lane;
        d2 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = p[k * 64];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s += r[k].x + r[k].y; q += r[k].x * r[k].x + r[k].y * r[k].y; }
    }
    const u64 t1 = now();
    s = wsum(s); q = wsum(q);
    if (lane == 0) { lds[w][0] = s; lds[w][1] = q; }
    unsigned old = 0;
    if (lane == 0) old = __hip_atomic_fetch_add(&lds_cnt, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
    old = __builtin_amdgcn_readfirstlane(old);
    const unsigned nw = WAVES - ((has_monitor && blockIdx.x == 0) ? 1u : 0u);
    if (old + 1 == nw) {   // the workgroup's last wave sums the waves and publishes
        double ws = 0, wq = 0;
        if (lane < WAVES && !(has_monitor && blockIdx.x == 0 && lane == 0)) { ws = lds[lane][0]; wq = lds[lane][1]; }
        ws = wsum(ws); wq = wsum(wq);
        // seven values per workgroup, as the product carries: (n, S, Q) x 2 groups + visited
        const double val[7] = {ws, wq, 0.5 * ws, 0.25 * wq, 1.0, 2.0, 3.0};
        double* mine = a.partial + (size_t)blockIdx.x * 8;
        if (a.mode == 4) {
            if (lane < 4) {
                const double pv = lane == 0 ? val[0] : lane == 1 ? val[1] : lane == 2 ? val[2] : val[3];
                const unsigned cw = lane == 0 ? 1u : lane == 1 ? 2u : lane == 2 ? 3u : 0u;
                store_granule(reinterpret_cast<char*>(mine) + lane * 16, pv, cw, a.epoch + 1);
            }
        } else {
            if (lane < 7) {
                double pv = 0;
#pragma unroll
                for (int cc = 0; cc < 7; ++cc) pv = lane == cc ? val[cc] : pv;
                __hip_atomic_store(mine + lane, pv, RLX);
            }
            if (a.mode != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (a.mode == 3) {
                if (lane == 0) __hip_atomic_store(reinterpret_cast<u64*>(mine + 7), (u64)a.epoch + 1, RLX);
            } else if (a.mode == 1 || a.mode == 2) {
                unsigned is_last = 0;
                if (lane == 0) {
                    if (a.mode == 2) {
                        is_last = __hip_atomic_fetch_add(a.tickets + 16 * 32, 1u, RLX) == a.epoch * gridDim.x + gridDim.x - 1;
                    } else {
                        const unsigned shard = blockIdx.x & 7, per_shard = (gridDim.x + 7 - shard) / 8;
                        if (__hip_atomic_fetch_add(a.tickets + shard * 32, 1u, RLX) == a.epoch * per_shard + per_shard - 1)
                            is_last = __hip_atomic_fetch_add(a.tickets + 16 * 32, 1u, RLX) == a.epoch * 8 + 7;
                    }
                }
                is_last = __shfl(is_last, 0, 64);
                if (is_last) {
                    const unsigned c = lane & 7, j = lane >> 3;
                    double run = 0;
                    double x[32];
#pragma unroll
                    for (int m = 0; m < 32; ++m) { const unsigned g = j + 8 * m; x[m] = g < gridDim.x ? __hip_atomic_load(a.partial + (size_t)g * 8 + c, RLX) : 0.0; }
#pragma unroll
                    for (int m = 0; m < 32; ++m) run += x[m];
                    if (c == 7) run = 0;
                    run += __shfl_xor(run, 8, 64); run += __shfl_xor(run, 16, 64); run += __shfl_xor(run, 32, 64);
                    if (lane < 7) __hip_atomic_store(a.result + lane, run, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    if (lane == 0) __hip_atomic_store(a.result + 7, (double)a.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    }
    if (lane == 0) { u64* st = a.stamps + 3 * wid; st[0] = t0; st[1] = t1; st[2] = now(); }
}

__global__ void k_empty(u64* stamps) { if (threadIdx.x == 0 && blockIdx.x == 0) stamps[0] = now(); }

int main(int argc, char** argv) {
    const u64 N = 200000000;  // 1.6 GB column: sweeps below take windows of it
    double* x; CK(hipMalloc(&x, N * 8));
    {
        std::vector<double> h(1 << 22);
        for (size_t i = 0; i < h.size(); ++i) h[i] = 1.0 + (double)(i % 997);
        for (u64 o = 0; o < N; o += h.size()) CK(hipMemcpy(x + o, h.data(), std::min<u64>(h.size(), N - o) * 8, hipMemcpyHostToDevice));
    }
    double* partial; CK(hipMalloc(&partial, 64 * 4096)); CK(hipMemset(partial, 0, 64 * 4096));
    unsigned* tickets; CK(hipMalloc(&tickets, 32 * 32 * 4));
    u64* stamps; CK(hipMalloc(&stamps, 3 * 8 * 65536 + 64));
    double* result; CK(hipHostMalloc(&result, 64, hipHostMallocDefault));
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<u64> hs(3 * 65536 + 8);

    auto run = [&](const char* name, int waves, int grid, double mb, int mode, u64 offset_rows) -> int {
        Args a{};
        a.x = (const d2*)(x + offset_rows); a.tiles = (u64)(mb * 1e6 / 8192.0);
        a.partial = partial; a.tickets = tickets; a.result = result; a.stamps = stamps; a.mode = mode;
        CK(hipMemsetAsync(tickets, 0, 32 * 32 * 4, st));
        CK(hipMemsetAsync(partial, 0, 64 * 4096, st));
        CK(hipStreamSynchronize(st));
        const int R = 60;
        std::vector<float> us; std::vector<double> lastStart, firstDone, lastDone, lastEnd, seen, folded, stored;
        double expect_s = 0; bool bad = false;
        for (int i = 0; i < R; ++i) {
            a.epoch = (unsigned)i;
            if (waves == 16) hipExtLaunchKernelGGL(k_sweep<16>, dim3(grid), dim3(1024), 0, st, e0, e1, 0, a);
            else hipExtLaunchKernelGGL(k_sweep<4>, dim3(grid), dim3(256), 0, st, e0, e1, 0, a);
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (mode != 0) {
                if (i == 0) expect_s = result[0];
                if (result[0] != expect_s || result[7] != (double)i || result[4] != (double)grid) bad = true;
            }
            if (i < 10) continue;
            us.push_back(ms * 1e3f);
            const size_t V = (size_t)grid * waves;
            CK(hipMemcpy(hs.data(), stamps, V * 24 + 24, hipMemcpyDeviceToHost));
            u64 s0 = ~0ull, sl = 0, d0 = ~0ull, dl = 0, el = 0;
            for (size_t k = (mode >= 3 ? 1 : 0); k < V; ++k) { s0 = std::min(s0, hs[3 * k]); sl = std::max(sl, hs[3 * k]); d0 = std::min(d0, hs[3 * k + 1]); dl = std::max(dl, hs[3 * k + 1]); el = std::max(el, hs[3 * k + 2]); }
            lastStart.push_back((sl - s0) * 0.01); firstDone.push_back((d0 - s0) * 0.01); lastDone.push_back((dl - s0) * 0.01); lastEnd.push_back((el - s0) * 0.01);
            if (mode >= 3) { seen.push_back((hs[3 * V] - s0) * 0.01); folded.push_back((hs[3 * V + 1] - s0) * 0.01); stored.push_back((hs[3 * V + 2] - s0) * 0.01); }
        }
        auto med = [](auto v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        auto mn = [](auto v) { return *std::min_element(v.begin(), v.end()); };
        printf("%-30s grid %4dx%-2d %6.1f MB  launch med %6.2f min %6.2f us | last start %5.2f  loads done %5.2f..%5.2f  last sweeper end %5.2f",
               name, grid, waves, mb, med(us), mn(us), med(lastStart), med(firstDone), med(lastDone), med(lastEnd));
        if (mode >= 3) printf("  monitor: seen %5.2f folded %5.2f stored %5.2f", med(seen), med(folded), med(stored));
        printf("%s\n", bad ? "  RESULT MISMATCH" : "");
        fflush(stdout);
        return 0;
    };

    {
        std::vector<float> us;
        for (int i = 0; i < 60; ++i) {
            hipExtLaunchKernelGGL(k_empty, dim3(256), dim3(1024), 0, st, e0, e1, 0, stamps);
            CK(hipStreamSynchronize(st));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (i >= 10) us.push_back(ms * 1e3f);
        }
        std::sort(us.begin(), us.end());
        printf("empty kernel 256x1024: launch med %.2f min %.2f us\n", us[us.size() / 2], us[0]);
    }
    const char* names[5] = {"no hand-off", "sharded tickets", "one ticket counter", "flags + monitor", "tagged granules + monitor"};
    for (u64 off : {0ull, 100000000ull}) {
        printf("-- window at row %llu\n", off);
        for (double mb : {0.8, 32.0, 80.0, 160.0}) {
            for (int mode = 0; mode < 5; ++mode) if (run(names[mode], 16, 256, mb, mode, off)) return 1;
            for (int mode = 1; mode < 5; ++mode) if (run(names[mode], 16, 128, mb, mode, off)) return 1;
            for (int mode = 1; mode < 3; ++mode) if (run(names[mode], 4, 1024, mb, mode, off)) return 1;
            for (int mode = 1; mode < 3; ++mode) if (run(names[mode], 4, 2048, mb, mode, off)) return 1;
        }
    }
    return 0;
}
