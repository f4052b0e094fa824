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
// Differences here: general beta; the analysis runs with the reference's (THREADS 256, R 2, MIN_NNZ 1024) instance
// so a block has at most 256 rows and -- typically -- 1024 + one row of non-zeros: one 2048-product round with one
// vector of lanes per row, the lanes-per-row chosen per block from its row count; a plan-time 16-B digest per
// block replaces the kernel's dependent bp -> first_block_of_row -> rowptr loads;
// long-row slices write one partial per block and a fix-up kernel adds them in block order (deterministic, no
// atomics); the slice index is derived from the break-point table itself, so a long row that shares its
// first block with leading empty rows keeps its first slice (the reference computes the slice from
// first_block_of_row alone and skips it).
#include "device_utils.hpp"
#include "kernels.hpp"
#include "tile_stage.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

constexpr int kPlusNpt = kNnzPerThread;       // 2048-product tile: a typical block (MIN_NNZ 1024 + one row) needs ONE round
constexpr int kPlusTile = kThreads * kPlusNpt;
constexpr int kPlusMaxRows = kPlusThreads;     // the analysis never gives a block more rows than this (= kThreads)

// first block whose break point is row r, given the analysis' first_block_of_row entry for r
__device__ __forceinline__ int first_block_at_row(const int *__restrict__ bp, int fbr_r, int r) {
  const int f = fbr_r >> 1;
  return (bp[f] == r) ? f : f + 1;
}

// Plan-time digest of (break_points, first_block_of_row, rowptr): one 16-B record per block so the SpMV
// kernel starts streaming after ONE scalar load instead of the bp -> first_block_of_row -> rowptr chain.
//   normal block : {row_begin, row_end, nnz_begin, nnz_end}
//   long-row slice: {row, -1, slice_begin, slice_end}
__global__ __launch_bounds__(256) void plus_digest_kernel(int m, int nblocks, int long_chunk, const int *__restrict__ bp,
                                                          const int *__restrict__ fbr, const int *__restrict__ rp,
                                                          int4v *__restrict__ blk, int *__restrict__ has_long) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= nblocks) return;
  const int row_begin = bp[g];
  int row_end = bp[g + 1];
  row_end = row_end < m ? row_end : m;
  const int flag = fbr[row_begin];
  int4v rec;
  if ((flag & 1) == 0) {
    rec.x = row_begin;
    rec.y = row_end;
    rec.z = rp[row_begin];
    rec.w = rp[row_end];
  } else {
    const int r = row_begin;
    const int idx = g - first_block_at_row(bp, flag, r);
    const int c0 = rp[r] + idx * long_chunk;
    const bool last = bp[g + 1] != r;
    rec.x = r;
    rec.y = -1;
    rec.z = c0;
    rec.w = last ? rp[r + 1] : c0 + long_chunk;
    *has_long = 1; // idempotent store
  }
  blk[g] = rec;
}

template <bool NTC, bool NTV, bool HINT>
__global__ __launch_bounds__(kThreads) void plus_kernel(int nnz, int nblocks, int xcd_chunk, double alpha, double beta,
                                                        const int4v *__restrict__ blk, const int *__restrict__ rp,
                                                        const int *__restrict__ ci, const double *__restrict__ v,
                                                        const double *__restrict__ x, double *y, const double *yin,
                                                        double *__restrict__ partial, int m,
                                                        const int *__restrict__ guard, int *__restrict__ stale, int reverse,
                                                        const unsigned char *__restrict__ cold) {
  check_plan_guard(rp, m, guard, stale);
  __shared__ __attribute__((aligned(16))) double lds[kPlusTile]; // written 16 B at a time
  __shared__ double row_acc[kPlusMaxRows];
  __shared__ TileSpans spans;
  if (threadIdx.x == 0) spans.n = 0; // published by the barrier that follows the first staging
  int g = (reverse & 1) ? zigzag_block(blockIdx.x, nblocks) : static_cast<int>(blockIdx.x); // zigzag (dispatch.cpp)
  if (xcd_chunk > 0) g = xcd_chunked_block(g, nblocks, xcd_chunk);
  const int4v rec = blk[g]; // wave-uniform: one scalar 16-B load
  const int row_begin = rec.x;

  if (rec.y >= 0) {
    // ---- normal block: rows [row_begin, row_end), at most kThreads of them ----
    const int nrows = rec.y - row_begin;
    const int s0 = rec.z;
    const int s1 = rec.w;
    const int a0 = s0 & ~3;
    int w = 64; // lanes per row: as many as the block's row count leaves room for
    while (w > 1 && nrows * w > kThreads) w >>= 1;
    const int lane = threadIdx.x & (w - 1);
    const int vec_id = threadIdx.x / w;
    const int row = row_begin + vec_id;
    const bool live = vec_id < nrows;
    int r0 = 0, r1 = 0;
    if (live) {
      r0 = rp[row];
      r1 = rp[row + 1];
    }
    double acc = 0.0;
    for (int off = a0; off < s1; off += kPlusTile) {
      stage_products<kThreads, kPlusNpt, NTC, NTV, HINT>(lds, off, s1, nnz, ci, v, x, true, cold, (reverse & 2) != 0);
      __syncthreads();
      const int lo = (r0 > off ? r0 : off) - off;
      const int hi = (r1 < off + kPlusTile ? r1 : off + kPlusTile) - off;
      acc += tile_row_sum<kThreads>(lds, spans, lo, hi > lo ? hi : lo, lane, w); // long spans go to whole waves
      if (off + kPlusTile < s1) __syncthreads(); // the next round overwrites the tile
    }
    acc = group_sum_dyn(acc, w);
    if (live && lane == 0 && !(r1 == r0 && keeps_y(y, yin, beta))) store_y(y, yin, row, alpha, beta, acc); // (empty row, unchanged y: skipped)
  } else {
    // ---- slice [rec.z, rec.w) of the long row `row_begin` ----
    double s = 0.0;
    const int j0 = rec.z, j1 = rec.w;
    {
      // 4 consecutive non-zeros per lane per step (one 16-B colindex load, two 16-B value loads), all steps of the slice
      // issued back to back; the slice start is aligned down to a multiple of 4 and the (at most 3 + 3) foreign elements
      // at its ends are masked out of the sum.  (One 4-/8-byte load per lane per step ran a matrix made of long rows only
      // at 4.9 TB/s against 6.0-6.4 for the tile kernels.)
      const XGather xr = make_xgather(x, HINT);
      for (int base = (j0 & ~3) + 4 * static_cast<int>(threadIdx.x); base < j1; base += 4 * kThreads) {
        if (base + 4 <= nnz) {
          const int4v c = load_stream_i4<NTC>(ci + base);
          const double2v a0 = load_stream_d2<NTV>(v + base);
          const double2v a1 = load_stream_d2<NTV>(v + base + 2);
          double p0, p1, p2, p3;
          if (HINT) {
            const unsigned nib = static_cast<unsigned>(cold[base >> 3]) >> (base & 4);
            p0 = a0.x * gather_hinted(xr, c.x, nib & 1u);
            p1 = a0.y * gather_hinted(xr, c.y, nib & 2u);
            p2 = a1.x * gather_hinted(xr, c.z, nib & 4u);
            p3 = a1.y * gather_hinted(xr, c.w, nib & 8u);
          } else {
            p0 = a0.x * x[c.x], p1 = a0.y * x[c.y], p2 = a1.x * x[c.z], p3 = a1.y * x[c.w];
          }
          s += (base + 0 >= j0 && base + 0 < j1) ? p0 : 0.0;
          s += (base + 1 >= j0 && base + 1 < j1) ? p1 : 0.0;
          s += (base + 2 >= j0 && base + 2 < j1) ? p2 : 0.0;
          s += (base + 3 >= j0 && base + 3 < j1) ? p3 : 0.0;
        } else {
          for (int e = 0; e < 4; ++e)
            if (base + e >= j0 && base + e < j1) s += v[base + e] * x[ci[base + e]];
        }
      }
    }
    s = group_sum<64>(s);
    constexpr int kWaves = kThreads / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) row_acc[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double total = 0.0;
#pragma unroll
      for (int i = 0; i < kWaves; ++i) total += row_acc[i];
      partial[g] = total;
    }
  }
}

// One thread per block; the first slice of each long row adds the row's slices in block order.
__global__ __launch_bounds__(256) void plus_fixup_kernel(int m, int nblocks, double alpha, double beta,
                                                         const int *__restrict__ bp, const int *__restrict__ fbr,
                                                         const double *__restrict__ partial,
                                                         double *y, const double *yin) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  // no early return: every lane of a wave takes part in wave_range_sum
  int r = -1, k0 = 0, k1 = 0;
  if (g < nblocks) {
    const int row = bp[g];
    if (row < m) {
      const int flag = fbr[row];
      if ((flag & 1) != 0 && g == first_block_at_row(bp, flag, row)) {
        r = row;
        // the row's slices are the blocks [g, e) with bp[.] == row; bp is non-decreasing: e by binary search
        int lo = g + 1, hi = nblocks;
        while (lo < hi) {
          const int mid = lo + (hi - lo) / 2;
          if (bp[mid] == row) lo = mid + 1; else hi = mid;
        }
        k0 = g;
        k1 = lo;
      }
    }
  }
  const double s = wave_range_sum(partial, k0, k1);
  if (r >= 0) store_y(y, yin, r, alpha, beta, s);
}

} // namespace

void launch_plus_digest(hipStream_t stream, const CsrDev &A, const int *bp, const int *fbr, int nblocks, int long_chunk,
                        void *blk, int *d_has_long) {
  if (nblocks <= 0) return;
  SPMV_ACC_LAUNCH(plus_digest_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, stream, A.m, nblocks, long_chunk, bp, fbr, A.rp,
                     static_cast<int4v *>(blk), d_has_long);
}

void launch_plus(hipStream_t stream, const CsrDev &A, const int *bp, const int *fbr, const void *blk, int nblocks,
                 bool has_long_rows, int xcd_chunk, int stream_policy, double *partial, double alpha, double beta,
                 const double *x, double *y, bool reverse) {
  if (nblocks <= 0) return;
#define SPMV_ACC_LAUNCH_PLUS(NC, NV, H)                                                                             \
  SPMV_ACC_LAUNCH((plus_kernel<NC, NV, H>), dim3(nblocks), dim3(kThreads), 0, stream, A.nnz, nblocks, xcd_chunk, \
                     alpha, beta, static_cast<const int4v *>(blk), A.rp, A.ci, A.v, x, y, A.yin ? A.yin : y, partial, A.m, A.guard, A.stale, (reverse ? 1 : 0) | (x32_ok(A) ? 2 : 0), \
                     A.cold)
  if (A.cold != nullptr) { // gather hints (kernels.hpp): cold gathers non-temporal
    switch (stream_policy & 3) {
    case 1: SPMV_ACC_LAUNCH_PLUS(false, false, true); break;
    case 2: SPMV_ACC_LAUNCH_PLUS(false, true, true); break;
    case 3: SPMV_ACC_LAUNCH_PLUS(true, false, true); break;
    default: SPMV_ACC_LAUNCH_PLUS(true, true, true); break;
    }
  } else {
    switch (stream_policy & 3) {
    case 1: SPMV_ACC_LAUNCH_PLUS(false, false, false); break;
    case 2: SPMV_ACC_LAUNCH_PLUS(false, true, false); break;
    case 3: SPMV_ACC_LAUNCH_PLUS(true, false, false); break;
    default: SPMV_ACC_LAUNCH_PLUS(true, true, false); break;
    }
  }
#undef SPMV_ACC_LAUNCH_PLUS
  if (has_long_rows) {
    SPMV_ACC_LAUNCH(plus_fixup_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, stream, A.m, nblocks, alpha, beta,
                       bp, fbr, partial, y, A.yin ? A.yin : y);
  }
}

} // namespace spmv_acc
