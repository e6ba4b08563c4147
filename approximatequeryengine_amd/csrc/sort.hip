// sort.hip — ascending sort of the amount column with its row permutation, for stratified_block_sample
// (custom_bplus_db.cpp:1342-1345 sorts a copy of every record by amount).  A one-off pre-pass per table, not a
// hot kernel: it uses rocPRIM's device radix sort (ROCm's own primitives library) rather than a hand-written one.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include "kernels.hpp"

namespace aqe {

hipError_t sort_amounts(const double* amount, uint64_t n, double* sorted_amount, uint32_t* sorted_row, hipStream_t s) {
    if (n == 0) return hipSuccess;
    if (n > 0xFFFFFFFFull) return hipErrorInvalidValue;
    size_t bytes = 0;
    rocprim::counting_iterator<uint32_t> rows(0);
    hipError_t e = rocprim::radix_sort_pairs(nullptr, bytes, amount, sorted_amount, rows, sorted_row, n, 0, 64, s);
    if (e != hipSuccess) return e;
    void* tmp = nullptr;
    e = hipMalloc(&tmp, bytes);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(tmp, bytes, amount, sorted_amount, rows, sorted_row, n, 0, 64, s);
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    return e != hipSuccess ? e : e2;
}

namespace {

// One value per lane: the first position whose amount is >= v (rows below v) and the first whose amount is > v
// (rows at or below v), by bisection on the sorted column.
__global__ void k_sorted_counts(const double* __restrict__ sorted, uint64_t n, const double* __restrict__ values, uint32_t m,
                                uint64_t* __restrict__ n_less, uint64_t* __restrict__ n_less_equal) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const double v = values[i];
    uint64_t a = 0, b = n;
    while (a < b) {
        const uint64_t mid = a + (b - a) / 2;
        if (sorted[mid] < v) a = mid + 1; else b = mid;
    }
    n_less[i] = a;
    b = n;  // (the second bound is at or past the first)
    while (a < b) {
        const uint64_t mid = a + (b - a) / 2;
        if (sorted[mid] <= v) a = mid + 1; else b = mid;
    }
    n_less_equal[i] = a;
}

}  // namespace

// stratified_block_sample over a sharded table (aqe_sorted_counts): where a list of values falls in this shard's sorted column.
hipError_t sorted_counts(const double* sorted_amount, uint64_t n, const double* values, uint32_t m, uint64_t* n_less,
                         uint64_t* n_less_equal, hipStream_t s) {
    if (m == 0) return hipSuccess;
    if (n == 0) {
        std::memset(n_less, 0, m * sizeof(uint64_t));
        std::memset(n_less_equal, 0, m * sizeof(uint64_t));
        return hipSuccess;
    }
    char* d = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&d), static_cast<size_t>(m) * 24);
    if (e != hipSuccess) return e;
    double* d_val = reinterpret_cast<double*>(d);
    uint64_t* d_lt = reinterpret_cast<uint64_t*>(d + static_cast<size_t>(m) * 8);
    uint64_t* d_le = reinterpret_cast<uint64_t*>(d + static_cast<size_t>(m) * 16);
    e = hipMemcpyAsync(d_val, values, static_cast<size_t>(m) * 8, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(k_sorted_counts, dim3((m + 255) / 256), dim3(256), 0, s, sorted_amount, n, d_val, m, d_lt, d_le);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(n_less, d_lt, static_cast<size_t>(m) * 8, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(n_less_equal, d_le, static_cast<size_t>(m) * 8, hipMemcpyDeviceToHost, s);
    const hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(d);
    return e != hipSuccess ? e : e2;
}

}  // namespace aqe
