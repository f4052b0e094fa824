// k_plus.hip -- row-block-plus family: variable row blocks produced by the host form of the
// row-block preprocessing pass (csr-adaptive-plus analysis).
//
// Reference roles:
//   * hip-csr-adaptive-plus/csr_adaptive_plus_spmv_imp.inl:31-53 (line_enhance_plus_kernel): block g
//     owns rows [bp[g], bp[g+1]); the low bit of first_block_of_row[bp[g]] marks a block that holds a
//     slice of one very long row.
//   * :123-205 (line_enhance_plus): normal block = line-enhance body on the block's rows.
//   * :55-121 (line_enhance_plus_shared_block): long-row block = block-wide sum + atomicAdd(y, alpha*s)
//     with beta dropped (the comment at :112-116 says so).
// Differences here: general beta; long-row slices write one partial per block and a fix-up kernel
// adds them in block order (deterministic, no atomics); the slice index is derived from the
// break-point table itself, so a long row that shares its first block with leading empty rows keeps
// its first slice (the reference computes the slice from first_block_of_row alone and skips it).
#include "device_utils.hpp"
#include "kernels.hpp"
#include "tile_stage.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

constexpr int kPlusNpt = 8;
constexpr int kPlusTile = kPlusThreads * kPlusNpt; // 4096 products = 32 KB

// first block whose break point is row r, given the analysis' first_block_of_row entry for r
__device__ __forceinline__ int first_block_at_row(const int *__restrict__ bp, int fbr_r, int r) {
  const int f = fbr_r >> 1;
  return (bp[f] == r) ? f : f + 1;
}

template <int VEC, bool ALIGNED>
__global__ __launch_bounds__(kPlusThreads) void plus_kernel(int m, int nnz, double alpha, double beta,
                                                            const int *__restrict__ bp, const int *__restrict__ fbr,
                                                            const int *__restrict__ rp, const int *__restrict__ ci,
                                                            const double *__restrict__ v,
                                                            const double *__restrict__ x, double *__restrict__ y,
                                                            double *__restrict__ partial) {
  __shared__ double lds[kPlusTile];
  const int g = blockIdx.x;
  const int row_begin = bp[g];
  int row_end = bp[g + 1];
  row_end = row_end < m ? row_end : m;
  const int flag = fbr[row_begin];

  if ((flag & 1) == 0) {
    // ---- normal block: rows [row_begin, row_end), at most kPlusThreads / VEC of them ----
    const int s0 = rp[row_begin];
    const int s1 = rp[row_end];
    const int lane = threadIdx.x % VEC;
    const int row = row_begin + threadIdx.x / VEC;
    const bool live = row < row_end;
    int r0 = 0, r1 = 0;
    if (live) {
      r0 = rp[row];
      r1 = rp[row + 1];
    }
    double acc = 0.0;
    for (int off = s0 & ~3; off < s1; off += kPlusTile) {
      stage_products<kPlusThreads, kPlusNpt, ALIGNED>(lds, off, s1, nnz, ci, v, x);
      __syncthreads();
      const int lo = (r0 > off ? r0 : off) - off;
      const int hi = (r1 < off + kPlusTile ? r1 : off + kPlusTile) - off;
      for (int j = lo + lane; j < hi; j += VEC) acc += lds[j];
      if (off + kPlusTile < s1) __syncthreads();
    }
    acc = group_sum<VEC>(acc);
    if (live && lane == 0) store_y(y, row, alpha, beta, acc);
  } else {
    // ---- slice of the long row `row_begin` ----
    const int r = row_begin;
    const int idx = g - first_block_at_row(bp, flag, r);
    const int c0 = rp[r] + idx * kPlusLongChunk;
    const bool last = bp[g + 1] != r;
    const int c1 = last ? rp[r + 1] : c0 + kPlusLongChunk;
    double s = 0.0;
    for (int j = c0 + threadIdx.x; j < c1; j += kPlusThreads) {
      s += load_stream(v + j) * x[load_stream(ci + j)];
    }
    s = group_sum<64>(s);
    constexpr int kWaves = kPlusThreads / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) lds[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double total = 0.0;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) total += lds[w];
      partial[g] = total;
    }
  }
}

// One thread per block; the first slice of each long row adds the row's slices in block order.
__global__ __launch_bounds__(256) void plus_fixup_kernel(int m, int nblocks, double alpha, double beta,
                                                         const int *__restrict__ bp, const int *__restrict__ fbr,
                                                         const double *__restrict__ partial,
                                                         double *__restrict__ y) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nblocks) return;
  const int r = bp[g];
  if (r >= m) return;
  const int flag = fbr[r];
  if ((flag & 1) == 0) return;
  if (g != first_block_at_row(bp, flag, r)) return;
  double s = 0.0;
  for (int k = g; k < nblocks && bp[k] == r; ++k) s += partial[k];
  store_y(y, r, alpha, beta, s);
}

template <int VEC>
void launch_vec(hipStream_t stream, const CsrDev &A, const int *bp, const int *fbr, int nblocks, double *partial,
                double alpha, double beta, const double *x, double *y) {
  if (A.aligned16) {
    hipLaunchKernelGGL((plus_kernel<VEC, true>), dim3(nblocks), dim3(kPlusThreads), 0, stream, A.m, A.nnz, alpha, beta,
                       bp, fbr, A.rp, A.ci, A.v, x, y, partial);
  } else {
    hipLaunchKernelGGL((plus_kernel<VEC, false>), dim3(nblocks), dim3(kPlusThreads), 0, stream, A.m, A.nnz, alpha,
                       beta, bp, fbr, A.rp, A.ci, A.v, x, y, partial);
  }
}

} // namespace

void launch_plus(hipStream_t stream, const CsrDev &A, const int *bp, const int *fbr, int nblocks, int vec,
                 double *partial, double alpha, double beta, const double *x, double *y) {
  if (nblocks <= 0) return;
  switch (vec) {
  case 1: launch_vec<1>(stream, A, bp, fbr, nblocks, partial, alpha, beta, x, y); break;
  case 2: launch_vec<2>(stream, A, bp, fbr, nblocks, partial, alpha, beta, x, y); break;
  case 4: launch_vec<4>(stream, A, bp, fbr, nblocks, partial, alpha, beta, x, y); break;
  case 8: launch_vec<8>(stream, A, bp, fbr, nblocks, partial, alpha, beta, x, y); break;
  case 16: launch_vec<16>(stream, A, bp, fbr, nblocks, partial, alpha, beta, x, y); break;
  case 32: launch_vec<32>(stream, A, bp, fbr, nblocks, partial, alpha, beta, x, y); break;
  default: launch_vec<64>(stream, A, bp, fbr, nblocks, partial, alpha, beta, x, y); break;
  }
}

void launch_plus_fixup(hipStream_t stream, const CsrDev &A, const int *bp, const int *fbr, int nblocks,
                       const double *partial, double alpha, double beta, double *y) {
  if (nblocks <= 0) return;
  hipLaunchKernelGGL(plus_fixup_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, stream, A.m, nblocks, alpha, beta,
                     bp, fbr, partial, y);
}

} // namespace spmv_acc
