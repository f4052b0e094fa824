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
#include "tile_stage.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

// w lanes per row, w wave-uniform per workgroup.  blocks [0, nb0) serve rows [0, row_split) with
// width w0, the rest serve [row_split, m) with width w1.  A lane group owns kVecRows rows (kThreads / w apart, so that
// neighbouring groups read neighbouring rowptr entries) and asks for the row extents of all of them, then for the first
// two steps of all of them, before it consumes anything: 2 * kVecRows stream loads and as many gathers per lane in flight
// instead of one (one row per group measured 2.8 TB/s on the boneS10-like matrix: a chain of three dependent latencies
// with nothing beside it).
// (kVecRows = 1 for matrices whose rows are longer than the two hoisted steps: their tails would otherwise run one row
// after the other inside a group.)
template <int kVecRows>
__global__ __launch_bounds__(kThreads) void vector_row_kernel(int m, int row_split, int nb0, int w0, int w1,
                                                              double alpha, double beta,
                                                              const int *__restrict__ rp, const int *__restrict__ ci,
                                                              const double *__restrict__ v,
                                                              const double *__restrict__ x, double *y, const double *yin,
                                                              const int *__restrict__ guard, int *__restrict__ stale) {
  check_plan_guard(rp, m, guard, stale);
  const bool second = static_cast<int>(blockIdx.x) >= nb0;
  const int w = second ? w1 : w0;
  const int groups = kThreads / w;
  const int row_lo = second ? row_split : 0;
  const int row_hi = second ? m : row_split;
  const int b = second ? blockIdx.x - nb0 : blockIdx.x;
  const int lane = threadIdx.x & (w - 1);
  const long long first = static_cast<long long>(row_lo) + static_cast<long long>(b) * groups * kVecRows + threadIdx.x / w;

  int j0[kVecRows], j1[kVecRows];
  bool live[kVecRows];
#pragma unroll
  for (int k = 0; k < kVecRows; ++k) {
    const long long r = first + static_cast<long long>(k) * groups;
    live[k] = r < row_hi;
    j0[k] = j1[k] = 0;
    if (live[k]) {
      j0[k] = rp[r];
      j1[k] = rp[r + 1];
    }
  }
  // first two steps of every row: all loads, then all gathers
  int c[kVecRows][2];
  double a[kVecRows][2];
#pragma unroll
  for (int k = 0; k < kVecRows; ++k) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = j0[k] + lane + i * w;
      c[k][i] = -1;
      a[k][i] = 0.0;
      if (j < j1[k]) {
        c[k][i] = load_stream(ci + j);
        a[k][i] = load_stream(v + j);
      }
    }
  }
  double s[kVecRows];
#pragma unroll
  for (int k = 0; k < kVecRows; ++k) {
    s[k] = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i)
      if (c[k][i] >= 0) s[k] += a[k][i] * x[c[k][i]];
  }
  // rows longer than two steps: four steps at a time, loads before gathers
#pragma unroll
  for (int k = 0; k < kVecRows; ++k) {
    for (int j = j0[k] + lane + 2 * w; j < j1[k]; j += 4 * w) {
      int cc[4];
      double aa[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int jj = j + u * w;
        cc[u] = -1;
        aa[u] = 0.0;
        if (jj < j1[k]) {
          cc[u] = load_stream(ci + jj);
          aa[u] = load_stream(v + jj);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (cc[u] >= 0) s[k] += aa[u] * x[cc[u]];
    }
  }
#pragma unroll
  for (int k = 0; k < kVecRows; ++k) {
    const double sum = group_sum_dyn(s[k], w); // every lane takes part (DPP needs a full exec mask)
    if (live[k] && lane == 0) store_y(y, yin, static_cast<int>(first + static_cast<long long>(k) * groups), alpha, beta, sum);
  }
}

// CSR-Vector on the tile machinery (the form KERNEL_STRATEGY=VECTOR_ROW / LIGHT and the two-width split run by default): a
// workgroup owns `rpb` consecutive rows, their non-zeros are staged into the LDS tile with the 16-B stream loads shared by
// the other tile kernels, and every row is summed by w lanes -- one w per matrix half for the split
// (vector_row_adaptive.hpp:72-142).  A lane group owns up to kVecTileRows rows (vecs apart), so that w lanes per row and a full
// 2048-product tile go together.  The width rule is the reference's shape (a power of two from the average row length,
// vector_row.cpp:15-27) with another constant: the reference gives a lane ~2 non-zeros because its lanes stream from global
// memory; here the products already sit in LDS and a lane adds up to 8 of them (w = pow2 >= avg / 8, at least 2) -- with the
// reference's widths the butterflies of 16- and 32-lane groups dominated the kernel (boneS10-sized 96 us at w = 16 against
// 77 us at w = 4; Bump_2911-sized 264 against 195 us).
constexpr int kVecTileRows = 2;
template <bool NTC, bool NTV>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(7, 8))) void vector_tile_kernel(int m, int nnz, int row_split, int nb0, int w0, int w1, int rpb0,
                                                               int rpb1, int xcd_chunk, double alpha, double beta,
                                                               const int *__restrict__ rp, const int *__restrict__ ci,
                                                               const double *__restrict__ v, const double *__restrict__ x,
                                                               double *y, const double *yin, const int *__restrict__ guard,
                                                               int *__restrict__ stale, int reverse) {
  check_plan_guard(rp, m, guard, stale);
  __shared__ __attribute__((aligned(16))) double lds[kTile];
  __shared__ TileSpans spans;
  if (threadIdx.x == 0) spans.n = 0; // published by the barrier that follows the first staging
  const int nblocks = gridDim.x;
  int blk = (reverse & 1) ? zigzag_block(blockIdx.x, nblocks) : static_cast<int>(blockIdx.x); // zigzag (dispatch.cpp)
  if (xcd_chunk > 0) blk = xcd_chunked_block(blk, nblocks, xcd_chunk);
  const bool second = blk >= nb0; // block-uniform: which matrix half (one width when row_split == m)
  const int w = second ? w1 : w0;
  const int rpb = second ? rpb1 : rpb0;
  const int half_lo = second ? row_split : 0;
  const int half_hi = second ? m : row_split;
  const long long base_ll = static_cast<long long>(half_lo) + static_cast<long long>(second ? blk - nb0 : blk) * rpb;
  const int row_base = static_cast<int>(base_ll);
  const int row_end = base_ll + rpb < half_hi ? row_base + rpb : half_hi;
  const int s0 = rp[row_base];
  const int s1 = rp[row_end];
  const int lane = threadIdx.x & (w - 1);
  const int log2w = 31 - __builtin_clz(w); // w is a power of two
  const int vecs = kThreads >> log2w;
  const int vec_id = threadIdx.x >> log2w;
  int r0[kVecTileRows], r1[kVecTileRows];
#pragma unroll
  for (int k = 0; k < kVecTileRows; ++k) {
    const int row = row_base + vec_id + k * vecs;
    r0[k] = r1[k] = 0;
    if (row < row_end) {
      r0[k] = rp[row];
      r1[k] = rp[row + 1];
    }
  }
  double acc[kVecTileRows];
#pragma unroll
  for (int k = 0; k < kVecTileRows; ++k) acc[k] = 0.0;
  for (int off = s0 & ~3; off < s1; off += kTile) {
    stage_products<kThreads, kNnzPerThread, NTC, NTV>(lds, off, s1, nnz, ci, v, x, true, nullptr, (reverse & 2) != 0);
    __syncthreads();
    int lo[kVecTileRows], hi[kVecTileRows];
#pragma unroll
    for (int k = 0; k < kVecTileRows; ++k) {
      lo[k] = (r0[k] > off ? r0[k] : off) - off;
      const int h = (r1[k] < off + kTile ? r1[k] : off + kTile) - off;
      hi[k] = h > lo[k] ? h : lo[k];
    }
    tile_rows_sum<kThreads, kVecTileRows>(lds, spans, lo, hi, lane, w, acc);
    if (off + kTile < s1) __syncthreads(); // the next round overwrites the tile
  }
#pragma unroll
  for (int k = 0; k < kVecTileRows; ++k) {
    const double s = group_sum_dyn(acc[k], w); // every lane takes part
    const int row = row_base + vec_id + k * vecs;
    if (row < row_end && lane == 0) store_y(y, yin, row, alpha, beta, s);
  }
}

// One 64-lane wavefront per row, for rows of hundreds of non-zeros (reference role: hip-wf-row/, KERNEL_STRATEGY
// WF_ROW -- src/acc/hip-wf-row/spmv_hip.cpp:18-34, one 4/8-byte load per lane per step there).  A lane owns 4
// consecutive non-zeros per step (one 16-B colindex load, two 16-B value loads) and two steps are in flight, so a
// wave keeps 512 non-zeros = 6 KB of stream + 512 gathers outstanding; the row start is aligned down to a multiple
// of 4 and the (at most 3 + 3) foreign elements at the row's ends are masked out of the sum.
__global__ __launch_bounds__(kThreads) void wave_row_kernel(int m, int nnz, double alpha, double beta,
                                                            const int *__restrict__ rp, const int *__restrict__ ci,
                                                            const double *__restrict__ v,
                                                            const double *__restrict__ x, double *y, const double *yin,
                                                            const int *__restrict__ guard, int *__restrict__ stale) {
  check_plan_guard(rp, m, guard, stale);
  const int lane = threadIdx.x & (kWave - 1);
  // (one wavefront per row; beyond kMaxGridBlocks * 4 rows -- 33.5 M -- the wavefronts stride over the rows)
  for (long long row_ll = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave; row_ll < m;
       row_ll += static_cast<long long>(gridDim.x) * (kThreads / kWave)) {
  const bool live = true; // wave-uniform
  const int row = static_cast<int>(row_ll);
  double s = 0.0;
  if (live) {
    const int j0 = rp[row];
    const int j1 = rp[row + 1];
    {
      for (int base = (j0 & ~3) + 4 * lane; base < j1; base += 2 * 4 * kWave) {
        const int ia = base, ib = base + 4 * kWave;
        const bool fa = ia + 4 <= nnz;             // whole 16-B group inside the arrays
        const bool fb = ib < j1 && ib + 4 <= nnz;
        int4v ca, cb;
        double2v a0, a1, b0, b1;
        if (fa) {
          ca = load_stream_i4<true>(ci + ia);
          a0 = load_stream_d2<true>(v + ia);
          a1 = load_stream_d2<true>(v + ia + 2);
        }
        if (fb) {
          cb = load_stream_i4<true>(ci + ib);
          b0 = load_stream_d2<true>(v + ib);
          b1 = load_stream_d2<true>(v + ib + 2);
        }
        if (fa) {
          const double p0 = a0.x * x[ca.x], p1 = a0.y * x[ca.y], p2 = a1.x * x[ca.z], p3 = a1.y * x[ca.w];
          s += (ia + 0 >= j0 && ia + 0 < j1) ? p0 : 0.0;
          s += (ia + 1 >= j0 && ia + 1 < j1) ? p1 : 0.0;
          s += (ia + 2 >= j0 && ia + 2 < j1) ? p2 : 0.0;
          s += (ia + 3 >= j0 && ia + 3 < j1) ? p3 : 0.0;
        } else {
          for (int e = 0; e < 4; ++e)
            if (ia + e >= j0 && ia + e < j1) s += v[ia + e] * x[ci[ia + e]];
        }
        if (fb) {
          const double p0 = b0.x * x[cb.x], p1 = b0.y * x[cb.y], p2 = b1.x * x[cb.z], p3 = b1.y * x[cb.w];
          s += (ib + 0 < j1) ? p0 : 0.0;
          s += (ib + 1 < j1) ? p1 : 0.0;
          s += (ib + 2 < j1) ? p2 : 0.0;
          s += (ib + 3 < j1) ? p3 : 0.0;
        } else if (ib < j1) {
          for (int e = 0; e < 4; ++e)
            if (ib + e < j1) s += v[ib + e] * x[ci[ib + e]];
        }
      }
    }
  }
  s = group_sum<64>(s);
  if (live && lane == 0) store_y(y, yin, row, alpha, beta, s);
  }
}

// the stale-plan guard's samples: rowptr[k * m / 63], k = 0 .. 63 (same indices as device_utils.hpp::check_plan_guard)
__global__ __launch_bounds__(kWave) void guard_fill_kernel(const int *__restrict__ rp, int m, int *__restrict__ guard) {
  guard[threadIdx.x] = rp[static_cast<int>(static_cast<long long>(threadIdx.x) * m / (kWave - 1))];
}

// the guard check alone (one wavefront): for call paths whose SpMV kernels run on matrices DERIVED from the caller's (the slabs of
// the opt-in column-slab blocking), so that a structure edited in place is still noticed
__global__ __launch_bounds__(kWave) void guard_check_kernel(const int *__restrict__ rp, int m, const int *__restrict__ guard,
                                                            int *__restrict__ stale) {
  check_plan_guard(rp, m, guard, stale);
}

__global__ __launch_bounds__(kThreads) void scale_y_kernel(int m, double beta, double *y, const double *yin) {
  const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < m) y[i] = (beta == 0.0) ? 0.0 : beta * yin[i];
}

// Opt-in structural check of the caller's arrays (tunable `validate`): bit 0 rowptr decreases or is negative, bit 1
// rowptr[m] != nnz, bit 2 a column index outside [0, n).  Reads every index once; no kernel of the library is launched
// on a matrix that fails.
__global__ __launch_bounds__(kThreads) void validate_csr_kernel(const int *__restrict__ rp, const int *__restrict__ ci, int m,
                                                                int n, int nnz, int *__restrict__ flags) {
  const long long stride = static_cast<long long>(gridDim.x) * kThreads;
  int bad = 0;
  for (long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; i < m; i += stride) {
    const int a = rp[i], b = rp[i + 1];
    if (a < 0 || a > b) bad |= 1;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && rp[m] != nnz) bad |= 2;
  for (long long j = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; j < nnz; j += stride) {
    const int c = ci[j];
    if (c < 0 || c >= n) bad |= 4;
  }
  if (bad) atomicOr(flags, bad);
}

// streaming copy with the SpMV kernels' load shape (16 B per lane, non-temporal), 4 steps per lane
template <bool NT>
__global__ __launch_bounds__(kThreads) void stream_copy_kernel(int4v *__restrict__ dst, const int4v *__restrict__ src,
                                                               long long n16) {
  const long long base = (static_cast<long long>(blockIdx.x) * kThreads * 4) + threadIdx.x;
  int4v r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + static_cast<long long>(k) * kThreads;
    if (i < n16) r[k] = NT ? __builtin_nontemporal_load(src + i) : src[i];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const long long i = base + static_cast<long long>(k) * kThreads;
    if (i < n16) {
      if (NT) __builtin_nontemporal_store(r[k], dst + i);
      else dst[i] = r[k];
    }
  }
}

inline int ceil_div_ll(long long a, long long b) { return static_cast<int>((a + b - 1) / b); }

} // namespace

void launch_vector_row(hipStream_t stream, const CsrDev &A, int row_split, int w0, int w1, double alpha, double beta,
                       const double *x, double *y, bool single_row_groups) {
  if (A.m <= 0) return;
  if (row_split < 0) row_split = 0;
  if (row_split > A.m) row_split = A.m;
  const long long avg = static_cast<long long>(A.count()) / A.m;
  const int wide = w0 > w1 ? w0 : w1;
  // rows per lane group: 1 when rows outgrow the hoisted steps, or when the caller knows the row lengths are very uneven
  // (four hub rows of a power-law matrix in one group would run one after the other)
  const int rows = (single_row_groups || avg > 2LL * wide) ? 1 : 4;
  // (a forced width on a matrix of very many short rows would need more workgroups than a launch holds: narrower lane groups then)
  while ((static_cast<long long>(A.m) + rows * (kThreads / (w0 > w1 ? w0 : w1)) - 1) / (rows * (kThreads / (w0 > w1 ? w0 : w1))) + 1 > max_grid_blocks() &&
         (w0 > 1 || w1 > 1)) {
    if (w0 > 1) w0 >>= 1;
    if (w1 > 1) w1 >>= 1;
  }
  const int nb0 = ceil_div_ll(row_split, rows * (kThreads / w0));
  const int nb1 = ceil_div_ll(A.m - row_split, rows * (kThreads / w1));
  if (nb0 + nb1 == 0) return;
  if (rows == 1)
    SPMV_ACC_LAUNCH(vector_row_kernel<1>, dim3(nb0 + nb1), dim3(kThreads), 0, stream, A.m, row_split, nb0, w0, w1, alpha, beta,
                       A.rp, A.ci, A.v, x, y, A.yin ? A.yin : y, A.guard, A.stale);
  else
    SPMV_ACC_LAUNCH(vector_row_kernel<4>, dim3(nb0 + nb1), dim3(kThreads), 0, stream, A.m, row_split, nb0, w0, w1, alpha, beta,
                       A.rp, A.ci, A.v, x, y, A.yin ? A.yin : y, A.guard, A.stale);
}

void launch_vector_tile(hipStream_t stream, const CsrDev &A, int row_split, int w0, int w1, double avg0, double avg1,
                        int target_products, int xcd_chunk, int stream_policy, double alpha, double beta, const double *x,
                        double *y, bool reverse) {
  if (A.m <= 0) return;
  if (row_split < 0) row_split = 0;
  if (row_split > A.m) row_split = A.m;
  // rows per workgroup of a half: enough to fill the tile, at most kVecTileRows per lane group
  auto rows_for = [&](double avg, int w) {
    const long long cap = static_cast<long long>(kVecTileRows) * (kThreads / w);
    long long r = avg > 0.0 ? static_cast<long long>(target_products / avg) : cap;
    if (r > cap) r = cap;
    return static_cast<int>(r < 1 ? 1 : r);
  };
  const int rpb0 = rows_for(avg0, w0), rpb1 = rows_for(avg1, w1);
  const int nb0 = ceil_div_ll(row_split, rpb0);
  const int nb1 = ceil_div_ll(A.m - row_split, rpb1);
  if (nb0 + nb1 == 0) return;
#define SPMV_ACC_LAUNCH_VT(NC, NV)                                                                                    \
  SPMV_ACC_LAUNCH((vector_tile_kernel<NC, NV>), dim3(nb0 + nb1), dim3(kThreads), 0, stream, A.m, A.nnz, row_split, nb0, \
                     w0, w1, rpb0, rpb1, xcd_chunk, alpha, beta, A.rp, A.ci, A.v, x, y, A.yin ? A.yin : y, A.guard, A.stale, (reverse ? 1 : 0) | (x32_ok(A) ? 2 : 0))
  switch (stream_policy & 3) {
  case 1: SPMV_ACC_LAUNCH_VT(false, false); break;
  case 2: SPMV_ACC_LAUNCH_VT(false, true); break;
  case 3: SPMV_ACC_LAUNCH_VT(true, false); break;
  default: SPMV_ACC_LAUNCH_VT(true, true); break;
  }
#undef SPMV_ACC_LAUNCH_VT
}

void launch_wave_row(hipStream_t stream, const CsrDev &A, double alpha, double beta, const double *x, double *y) {
  if (A.m <= 0) return;
  int grid = ceil_div_ll(A.m, kThreads / kWave);
  if (grid > max_grid_blocks()) grid = max_grid_blocks();
  SPMV_ACC_LAUNCH(wave_row_kernel, dim3(grid), dim3(kThreads), 0, stream, A.m, A.nnz, alpha, beta, A.rp, A.ci, A.v,
                     x, y, A.yin ? A.yin : y, A.guard, A.stale);
}

void launch_stream_copy(hipStream_t stream, void *dst, const void *src, long long bytes, bool non_temporal) {
  const long long n16 = bytes / 16;
  if (n16 <= 0) return;
  const dim3 grid(ceil_div_ll(n16, kThreads * 4));
  if (non_temporal) {
    SPMV_ACC_LAUNCH(stream_copy_kernel<true>, grid, dim3(kThreads), 0, stream, static_cast<int4v *>(dst),
                       static_cast<const int4v *>(src), n16);
  } else {
    SPMV_ACC_LAUNCH(stream_copy_kernel<false>, grid, dim3(kThreads), 0, stream, static_cast<int4v *>(dst),
                       static_cast<const int4v *>(src), n16);
  }
}

void launch_validate_csr(hipStream_t stream, const CsrDev &A, int *d_flags) {
  const long long work = A.m > A.nnz ? A.m : A.nnz;
  long long blocks = (work + kThreads - 1) / kThreads;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  SPMV_ACC_LAUNCH(validate_csr_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, A.rp, A.ci, A.m, A.n,
                     A.nnz, d_flags);
}

void launch_guard_fill(hipStream_t stream, const int *rp, int m, int *d_guard) {
  static_assert(kGuardSamples == kWave, "one lane per sample");
  SPMV_ACC_LAUNCH(guard_fill_kernel, dim3(1), dim3(kWave), 0, stream, rp, m, d_guard);
}

void launch_guard_check(hipStream_t stream, const CsrDev &A) {
  if (A.guard && A.stale) SPMV_ACC_LAUNCH(guard_check_kernel, dim3(1), dim3(kWave), 0, stream, A.rp, A.m, A.guard, A.stale);
}

void launch_scale_y(hipStream_t stream, int m, double beta, double *y, const double *yin) {
  if (m <= 0) return;
  SPMV_ACC_LAUNCH(scale_y_kernel, dim3(ceil_div_ll(m, kThreads)), dim3(kThreads), 0, stream, m, beta, y, yin ? yin : y);
}

} // namespace spmv_acc
