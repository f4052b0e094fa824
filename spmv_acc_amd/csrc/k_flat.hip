// k_flat.hip -- FLAT family: non-zero splitting + the device form of the row-block preprocessing pass.
//
// Reference roles:
//   * hip-flat/flat_imp.inl:108-131 (pre_calc_break_point): break_points[j] = first row touched by
//     nnz block j.  The reference pre-zeroes the array (hipMalloc + hipMemset on EVERY SpMV call,
//     flat.cpp:39-40, never freed) and lets every row scatter into it.  Here each entry is computed
//     independently by one binary search over rowptr -- O(blocks * log m) reads instead of a full
//     pass over rowptr, no memset, no write conflicts -- and yields bit-identical values (including
//     the reference's conventions: bp[0] = 0, entries past the last block stay 0, a block that starts
//     exactly on a row boundary gets that row).
//   * hip-flat/flat_imp_one_pass.hpp:16-77 + flat_reduce.hpp (spmv_flat_one_pass_kernel): block b
//     multiplies nnz [b*S, (b+1)*S) into LDS and reduces per row.  The reference adds EVERY row's
//     result with an fp64 atomicAdd (so beta is ignored and y must hold the beta-term already).
//     Here rows that lie completely inside a tile are stored directly with the full
//     y = alpha*s + beta*y update; only the (at most two) rows cut by a tile edge produce a carry,
//     and a second tiny kernel folds the carries in a fixed order: general beta, no atomics,
//     bit-reproducible results.
#include "device_utils.hpp"
#include "kernels.hpp"
#include "tile_stage.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

// ---- preprocessing: break points -------------------------------------------------------------------
__global__ __launch_bounds__(256) void break_points_kernel(const int *__restrict__ rp, int m, int nnz, int stride,
                                                           int *__restrict__ bp, int bp_len) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= bp_len) return;
  const long long target = static_cast<long long>(j) * stride;
  int out = 0;
  if (j > 0 && target <= nnz) {
    // p = first index in [0, m] with rp[p] >= target   (rp[0] = 0 < target <= rp[m])
    int lo = 0, hi = m;
    while (lo < hi) {
      const int mid = lo + (hi - lo) / 2;
      if (rp[mid] < target) lo = mid + 1; else hi = mid;
    }
    out = (rp[lo] == target) ? lo : lo - 1;
  }
  bp[j] = out;
}

// ---- tile geometry helpers (device) ------------------------------------------------------------------
// Rows owned by tile t: [first, end_excl).  first = bp[t] (row containing nnz t*S, or the first row
// that starts there).  A row that starts exactly at the next tile's origin belongs to the next tile;
// the last tile owns every remaining row (trailing empty rows included).
__device__ __forceinline__ int tile_end_excl(const int *__restrict__ rp, const int *__restrict__ bp, int t, int ntiles,
                                             int m, int t1) {
  if (t == ntiles - 1) return m;
  int e = bp[t + 1];
  e = e < m ? e : m;
  return (e < m && rp[e] < t1) ? e + 1 : e;
}

template <bool ALIGNED>
__global__ __launch_bounds__(kThreads) void flat_tile_kernel(int m, int nnz, int ntiles, double alpha, double beta,
                                                             const int *__restrict__ rp, const int *__restrict__ bp,
                                                             const int *__restrict__ ci,
                                                             const double *__restrict__ v,
                                                             const double *__restrict__ x, double *__restrict__ y,
                                                             double *__restrict__ head, double *__restrict__ tail) {
  __shared__ double lds[kTile];
  const int t = blockIdx.x;
  const int t0 = t * kFlatStride; // host guarantees nnz + stride fits in int
  const int t1 = (nnz - t0 > kFlatStride) ? t0 + kFlatStride : nnz;

  stage_products<kThreads, kNnzPerThread, ALIGNED>(lds, t0, t1, nnz, ci, v, x);

  int first = bp[t];
  first = first < m ? first : m;
  const int end_excl = tile_end_excl(rp, bp, t, ntiles, m, t1);
  const int nrows = end_excl - first;

  // lanes per row for this tile: as many as the tile's row count leaves room for (wave-uniform)
  int w = 1;
  while (w < 64 && nrows * (w * 2) <= kThreads) w <<= 1;

  __syncthreads();

  const int lane = threadIdx.x & (w - 1);
  const int vec_id = threadIdx.x / w;
  const int vecs = kThreads / w;
  // all lanes walk the same number of iterations so the DPP reduction sees a full exec mask
  for (int base = 0; base < nrows; base += vecs) {
    const int r = first + base + vec_id;
    const bool live = (base + vec_id) < nrows;
    int a = 0, b = 0;
    if (live) {
      a = rp[r];
      b = rp[r + 1];
    }
    const int lo = (a > t0 ? a : t0) - t0;
    const int hi = (b < t1 ? b : t1) - t0;
    double s = 0.0;
    for (int j = lo + lane; j < hi; j += w) s += lds[j];
    s = group_sum_dyn(s, w);
    if (live && lane == 0) {
      if (a >= t0 && b <= t1) {
        store_y(y, r, alpha, beta, s); // complete row (possibly empty): final value
      } else if (a < t0) {
        head[t] = s; // row started in an earlier tile (it may also run past this one)
      } else {
        tail[t] = s; // row starts here and continues in the next tile
      }
    }
  }
}

// One thread per tile that holds the START of a cut row: adds its tail carry and the head carries of
// the following tiles in tile order, then applies alpha/beta once.
__global__ __launch_bounds__(256) void flat_fixup_kernel(int m, int nnz, int ntiles, double alpha, double beta,
                                                         const int *__restrict__ rp, const int *__restrict__ bp,
                                                         const double *__restrict__ head,
                                                         const double *__restrict__ tail, double *__restrict__ y) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles - 1) return; // the last tile cannot have a row that continues
  const int t0 = t * kFlatStride;
  const int t1 = t0 + kFlatStride; // t < ntiles-1, so the tile is full
  const int end_excl = tile_end_excl(rp, bp, t, ntiles, m, t1);
  int first = bp[t];
  first = first < m ? first : m;
  if (end_excl <= first) return;
  const int r = end_excl - 1;
  const int a = rp[r];
  const int b = rp[r + 1];
  if (!(b > t1 && a >= t0)) return; // row r does not start in this tile, or ends in it
  double s = tail[t];
  for (int k = t + 1; k < ntiles; ++k) {
    s += head[k];
    const long long k1 = static_cast<long long>(k + 1) * kFlatStride;
    if (b <= k1) break;
  }
  store_y(y, r, alpha, beta, s);
}

} // namespace

void launch_break_points(hipStream_t stream, const int *rp, int m, int nnz, int stride, int *bp, int bp_len) {
  if (bp_len <= 0) return;
  hipLaunchKernelGGL(break_points_kernel, dim3((bp_len + 255) / 256), dim3(256), 0, stream, rp, m, nnz, stride, bp,
                     bp_len);
}

void launch_flat(hipStream_t stream, const CsrDev &A, const int *bp, int ntiles, double *head, double *tail,
                 double alpha, double beta, const double *x, double *y) {
  if (ntiles <= 0) return;
  if (A.aligned16) {
    hipLaunchKernelGGL((flat_tile_kernel<true>), dim3(ntiles), dim3(kThreads), 0, stream, A.m, A.nnz, ntiles, alpha,
                       beta, A.rp, bp, A.ci, A.v, x, y, head, tail);
  } else {
    hipLaunchKernelGGL((flat_tile_kernel<false>), dim3(ntiles), dim3(kThreads), 0, stream, A.m, A.nnz, ntiles, alpha,
                       beta, A.rp, bp, A.ci, A.v, x, y, head, tail);
  }
}

void launch_flat_fixup(hipStream_t stream, const CsrDev &A, const int *bp, int ntiles, const double *head,
                       const double *tail, double alpha, double beta, double *y) {
  if (ntiles <= 1) return;
  hipLaunchKernelGGL(flat_fixup_kernel, dim3((ntiles - 1 + 255) / 256), dim3(256), 0, stream, A.m, A.nnz, ntiles,
                     alpha, beta, A.rp, bp, head, tail, y);
}

} // namespace spmv_acc
