// k_vector_row.hip -- DEFAULT strategy and the two-half vector-row split of ADAPTIVE.
//
// Reference roles:
//   * src/acc/hip/spmv_hip_acc_imp.cpp:15-35 (default_sparse_spmv): the semantic baseline
//     y = alpha*A*x + beta*y, run there by ONE GPU thread.  Here it is a real parallel kernel with
//     the same general alpha/beta semantics.
//   * src/acc/hip-vector-row/vector_row_adaptive.hpp:72-142 + vector_row.cpp:30-38
//     (adaptive_vec_row_sparse_spmv): the matrix is cut at m/2 and each half gets its own vector
//     width.  The reference splits a fixed 512-block grid's wavefronts 16 ways in proportion to the
//     halves' nnz; with a non-persistent grid (one workgroup per THREADS/w rows, thousands of
//     workgroups on 256 CUs) the hardware dispatcher does that balancing, so the split reduces to
//     "which width does this workgroup use".
#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

// w lanes per row, w wave-uniform per workgroup.  blocks [0, nb0) serve rows [0, row_split) with
// width w0, the rest serve [row_split, m) with width w1.
__global__ __launch_bounds__(kThreads) void vector_row_kernel(int m, int row_split, int nb0, int w0, int w1,
                                                              double alpha, double beta,
                                                              const int *__restrict__ rp, const int *__restrict__ ci,
                                                              const double *__restrict__ v,
                                                              const double *__restrict__ x, double *__restrict__ y) {
  const bool second = static_cast<int>(blockIdx.x) >= nb0;
  const int w = second ? w1 : w0;
  const int rows_per_block = kThreads / w;
  const int row_lo = second ? row_split : 0;
  const int row_hi = second ? m : row_split;
  const int b = second ? blockIdx.x - nb0 : blockIdx.x;
  const int lane = threadIdx.x & (w - 1);
  const long long row_ll = static_cast<long long>(row_lo) + static_cast<long long>(b) * rows_per_block + threadIdx.x / w;
  const bool live = row_ll < row_hi;
  const int row = static_cast<int>(row_ll);

  double s = 0.0;
  if (live) {
    const int j0 = rp[row];
    const int j1 = rp[row + 1];
    for (int j = j0 + lane; j < j1; j += w) {
      s += load_stream(v + j) * x[load_stream(ci + j)];
    }
  }
  s = group_sum_dyn(s, w); // every lane takes part (DPP needs a full exec mask)
  if (live && lane == 0) store_y(y, row, alpha, beta, s);
}

__global__ __launch_bounds__(kThreads) void scale_y_kernel(int m, double beta, double *__restrict__ y) {
  const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < m) y[i] = (beta == 0.0) ? 0.0 : beta * y[i];
}

// streaming copy with the SpMV kernels' load shape (16 B per lane, non-temporal), 4 steps per lane
__global__ __launch_bounds__(kThreads) void stream_copy_kernel(int4v *__restrict__ dst, const int4v *__restrict__ src,
                                                               long long n16) {
  const long long base = (static_cast<long long>(blockIdx.x) * kThreads * 4) + threadIdx.x;
  int4v r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + static_cast<long long>(k) * kThreads;
    if (i < n16) r[k] = __builtin_nontemporal_load(src + i);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + static_cast<long long>(k) * kThreads;
    if (i < n16) __builtin_nontemporal_store(r[k], dst + i);
  }
}

inline int ceil_div_ll(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

} // namespace

int pick_vec_width(int m, int nnz) {
  if (m <= 0) return 1;
  const long long avg = static_cast<long long>(nnz) / m;
  int w = 1;
  while (w < 64 && static_cast<long long>(w) * 8 < avg) w <<= 1;
  return w;
}

void launch_vector_row(hipStream_t stream, const CsrDev &A, int row_split, int w0, int w1, double alpha, double beta,
                       const double *x, double *y) {
  if (A.m <= 0) return;
  if (row_split < 0) row_split = 0;
  if (row_split > A.m) row_split = A.m;
  const int nb0 = ceil_div_ll(row_split, kThreads / w0);
  const int nb1 = ceil_div_ll(A.m - row_split, kThreads / w1);
  if (nb0 + nb1 == 0) return;
  hipLaunchKernelGGL(vector_row_kernel, dim3(nb0 + nb1), dim3(kThreads), 0, stream, A.m, row_split, nb0, w0, w1,
                     alpha, beta, A.rp, A.ci, A.v, x, y);
}

void launch_stream_copy(hipStream_t stream, void *dst, const void *src, long long bytes) {
  const long long n16 = bytes / 16;
  if (n16 <= 0) return;
  hipLaunchKernelGGL(stream_copy_kernel, dim3(ceil_div_ll(n16, kThreads * 4)), dim3(kThreads), 0, stream,
                     static_cast<int4v *>(dst), static_cast<const int4v *>(src), n16);
}

void launch_scale_y(hipStream_t stream, int m, double beta, double *y) {
  if (m <= 0) return;
  hipLaunchKernelGGL(scale_y_kernel, dim3(ceil_div_ll(m, kThreads)), dim3(kThreads), 0, stream, m, beta, y);
}

} // namespace spmv_acc
