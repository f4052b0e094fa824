// hip-flat/flat_compat.hpp -- what the reference benchmark's private copy of flat (benchmark/flat/spmv_acc_flat.cpp:20-71)
// takes from src/acc/hip-flat/spmv_hip_acc_imp.h besides the entry points: the two break-point kernels it launches itself and
// the two launch macros it expands.  Compiled only by hipcc (the kernels live in the including translation unit).
//   pre_calc_break_point<STRIDE, BLOCKS, I>     same table as flat_imp.inl:108-131 (bit-identical; written as one search per
//                                               entry, so it needs no pre-zeroed array and no particular launch shape)
//   pre_calc_break_point_v2<STRIDE, BLOCKS, I>  same table as flat_imp.inl:135-152 (entries for tiles that start inside the
//                                               non-zeros; the rest keep the caller's memset)
//   FLAT_KERNEL_WRAPPER / FLAT_KERNEL_ONE_PASS_WRAPPER(R, REDUCE_OPTION, REDUCE_VEC_SIZE, BLOCKS, THREADS)
//                                               expand, in the caller's scope (trans, alpha, beta, m, n, nnz, rowptr, colindex,
//                                               value, x, y), to this library's flat SpMV.  The library keeps its own break
//                                               points in the matrix's plan; the table the caller built is not read.
#ifndef SPMV_ACC_AMD_HIP_FLAT_FLAT_COMPAT_HPP
#define SPMV_ACC_AMD_HIP_FLAT_FLAT_COMPAT_HPP
#ifdef __HIPCC__

#include <hip/hip_runtime.h>

#include "../spmv_acc.h"

template <int BREAK_STRIDE, int BLOCKS, typename I>
__global__ void pre_calc_break_point(const I *__restrict__ row_ptr, const I m, I *__restrict__ break_points, const int bp_len) {
  const I nnz = row_ptr[m];
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; j < bp_len; j += step) {
    const long long target = j * BREAK_STRIDE;
    I out = 0;
    if (j > 0 && target <= nnz) {
      I lo = 0, hi = m; // first p in [0, m] with row_ptr[p] >= target
      while (lo < hi) {
        const I mid = lo + (hi - lo) / 2;
        if (row_ptr[mid] < target) lo = mid + 1; else hi = mid;
      }
      out = (row_ptr[lo] == target) ? lo : (lo > 0 ? lo - 1 : 0);
    }
    break_points[j] = out;
  }
}

template <int BREAK_STRIDE, int BLOCKS, typename I>
__global__ void pre_calc_break_point_v2(const I *__restrict__ row_ptr, const I m, I *__restrict__ break_points,
                                        const int bp_len) {
  const I nnz = row_ptr[m];
  const long long step = static_cast<long long>(gridDim.x) * blockDim.x;
  for (long long j = static_cast<long long>(blockIdx.x) * blockDim.x + threadIdx.x; j < bp_len; j += step) {
    const long long target = j * BREAK_STRIDE;
    if (target >= nnz) continue; // no row holds this non-zero: the entry keeps the caller's value
    I lo = 0, hi = m;            // first p in [0, m] with row_ptr[p] > target; the row that holds `target` is p - 1
    while (lo < hi) {
      const I mid = lo + (hi - lo) / 2;
      if (row_ptr[mid] <= target) lo = mid + 1; else hi = mid;
    }
    break_points[j] = lo - 1;
  }
}

#define FLAT_KERNEL_WRAPPER(R, REDUCE_OPTION, REDUCE_VEC_SIZE, BLOCKS, THREADS)                                        \
  spmv_acc_csr_spmv_strategy(SPMV_ACC_FLAT, trans, alpha, beta, m, n, nnz, nullptr, rowptr, colindex, value, x, y)
#define FLAT_KERNEL_ONE_PASS_WRAPPER(R, REDUCE_OPTION, REDUCE_VEC_SIZE, BLOCKS, THREADS)                               \
  spmv_acc_csr_spmv_strategy(SPMV_ACC_FLAT, trans, alpha, beta, m, n, nnz, nullptr, rowptr, colindex, value, x, y)

#endif // __HIPCC__
#endif
