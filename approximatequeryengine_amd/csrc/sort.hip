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

}  // namespace aqe
