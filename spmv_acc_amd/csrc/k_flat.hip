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
// tile0 (round 5): the table starts at absolute tile `tile0` (an un-rebased row sub-range whose first non-zero lies there): bp[j] is the break
// point of tile j + tile0; entry 0 stays 0 (row 0 of the view) like the reference's bp[0].
__global__ __launch_bounds__(256) void break_points_kernel(const int *__restrict__ rp, int m, int nnz, int stride, int tile0,
                                                           int *__restrict__ bp, int bp_len) {
  const int j = blockIdx.x * 256 + threadIdx.x;
  if (j >= bp_len) return;
  const long long target = (static_cast<long long>(j) + tile0) * stride;
  int out = 0;
  if (j > 0 && target <= nnz) {
    // p = first index in [0, m] with rp[p] >= target   (rp[0] = 0 < target <= rp[m])
    int lo = 0, hi = m;
    while (lo < hi) {
      const int mid = lo + (hi - lo) / 2;
      if (rp[mid] < target) lo = mid + 1; else hi = mid;
    }
    // (lo == 0 only when rowptr[0] > 0 -- a row shard passed without rebasing: every tile before its first non-zero
    // then points at row 0 and owns no rows)
    out = (rp[lo] == target) ? lo : (lo > 0 ? lo - 1 : 0);
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

// (amdgpu_waves_per_eu(7): the register allocator is asked to stay within 72 VGPRs -- 7 workgroups per CU -- where the
// plain kernel with the finishing prefetch would take 78; the stream-first variants need more registers by design)
// SEGSUM: the reference's other reduction of a flat tile (FLAT_SEGMENT_SUM_REDUCE, hip-flat/flat.cpp:59-76 +
// common/utils.h:75-94 block_segment_sum: a Hillis-Steele segmented scan over the tile keyed by row index).  Here: row starts
// are flagged in LDS, every lane scans its NPT consecutive products in place (restarting at a flag), the lanes' open sums cross
// by a segmented scan over the 256 lanes (shuffles inside a wave, LDS across the four waves), and a row's sum is the scanned
// value at its last product.  No lane-group width, no dependence on the tile's row-length mix; one lane per row afterwards.
// HINT: gather hints (k_hint.hip): the tile's gathers take their cache policy from the plan's cold bits.
template <int NPT, bool NTC, bool NTV, bool EARLY, bool C16 = false, bool SEGSUM = false, bool HINT = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(EARLY || NPT > 8 ? 1 : 7, 8))) void flat_tile_kernel(int m, int nnz, int ntiles, double alpha, double beta,
                                                             const int *__restrict__ rp, const int *__restrict__ bp,
                                                             const int *__restrict__ ci,
                                                             const double *__restrict__ v,
                                                             const double *__restrict__ x, double *y, const double *yin,
                                                             double *__restrict__ head, double *__restrict__ tail,
                                                             int *__restrict__ tail_row, int *__restrict__ tail_end,
                                                             int xcd_chunk, int reach, int tile0,
                                                             const int *__restrict__ guard, int *__restrict__ stale,
                                                             Col16Dev c16, int reverse, int cache_ends,
                                                             const int4v *__restrict__ dig, const unsigned char *__restrict__ cold) {
  check_plan_guard(rp, m, guard, stale);
  // reach: a tile finishes its last row itself when it ends at most `reach` (0 or kFlatFinish) non-zeros past the tile
  constexpr int STRIDE = kThreads * NPT;
  static_assert(STRIDE / 64 <= kTileSpans, "TileSpans must hold every >= 64-product span of one tile");
  __shared__ __attribute__((aligned(16))) double lds[STRIDE]; // written 16 B at a time
  __shared__ double sh_tail_sum, sh_tail_yold;                 // partial (and old y) of the row this tile will finish itself
  __shared__ int sh_tail_row;
  __shared__ TileSpans spans;
  if (threadIdx.x == 0) {
    sh_tail_row = -1;
    spans.n = 0;
  }
  int t = (reverse & 1) ? zigzag_block(blockIdx.x, ntiles) : static_cast<int>(blockIdx.x); // every other SpMV walks the tiles backwards
  if (xcd_chunk > 0) t = xcd_chunked_block(t, ntiles, xcd_chunk);
  const int t0 = (t + tile0) * STRIDE; // host guarantees nnz + stride fits in int
  const int t1 = (nnz - t0 > STRIDE) ? t0 + STRIDE : nnz;

  // EARLY (chosen by timing on small grids, tuner.cpp): the tile's stream loads go out before anything else, so the
  // break point -> rowptr chain below overlaps them instead of preceding them.
  StreamRegs<EARLY ? NPT : 4> early_regs;
  const bool early = EARLY && xcd_chunk >= 0 && stage_fast_ok(t1, nnz); // (block-uniform)
  if (EARLY && early) stage_issue<kThreads, EARLY ? NPT : 4, NTC, NTV>(early_regs, t0, t1, ci, v);

  // The tile's row range first (scalar chain bp -> rowptr), then each lane group's row extents and old y, THEN the
  // stream: the reduction after the barrier finds everything in registers (no global latency follows the barrier on the
  // common path).  Issuing the stream loads before this chain instead (split staging) was measured: +46 VGPRs, occupancy
  // 8 -> 5, 2-4 % slower on six of seven stand-ins.
  // (round 2: one 16-B record per tile, built once per plan -- {first row, end row, rowptr of the last row, rowptr past it} --
  // replaces the bp[t] -> bp[t+1] -> rowptr[.] chain and the two scalar loads of the finishing prefetch: one scalar load
  // instead of up to five dependent ones before the row extents can be requested)
  const int4v rec = dig[t];
  const int first = rec.x;
  const int end_excl = rec.y;
  const int nrows = end_excl - first;
  // lanes per row for this tile: as many as the tile's row count leaves room for (wave-uniform); one with the segmented scan
  int w = 1;
  while (!SEGSUM && w < 64 && nrows * (w * 2) <= kThreads) w <<= 1;
  const int lane = threadIdx.x & (w - 1);
  const int vec_id = threadIdx.x / w;
  const int vecs = kThreads / w;
  // row extents and old y of the first pass over the rows (the only pass unless the tile holds more than 256 rows)
  const bool live0 = vec_id < nrows;
  int a0 = 0, b0 = 0;
  if (live0) {
    a0 = rp[first + vec_id];
    b0 = rp[first + vec_id + 1];
  }
  const bool early_y = beta != 0.0;
  double y_old0 = 0.0;
  if (early_y && live0 && lane == 0) y_old0 = yin[first + vec_id]; // read for cut rows too (<= 2 per tile): harmless

  // Finish mode (reach > 0): the tile's last row, when it starts here and runs at most `reach` non-zeros past the tile, is
  // completed by this tile's first wave.  Its overhang [t1, row end) is requested NOW, next to the stream loads -- the row's
  // extents are two scalar loads away from end_excl -- so that nothing but a wave reduction is left for the end of the
  // workgroup.  (Reading the overhang after the row loop made every workgroup end on a chain of dependent loads, which is
  // why carries + the fix-up launch won on 150 us kernels; with the loads up front finishing is the cheaper form wherever it
  // is legal.)
  constexpr int FIN_STEPS = kFlatFinish / kWave;
  int fin_c[FIN_STEPS];
  double fin_v[FIN_STEPS];
  bool fin = false; // wave-uniform
  if (reach > 0 && nrows > 0 && t < ntiles - 1 && threadIdx.x < kWave) {
    const int la = rec.z, lb = rec.w;
    fin = la >= t0 && lb > t1 && lb - t1 <= reach;
    if (fin) {
#pragma unroll
      for (int k = 0; k < FIN_STEPS; ++k) {
        const int j = t1 + static_cast<int>(threadIdx.x) + k * kWave;
        fin_c[k] = j < lb ? load_stream(ci + j) : -1;
        fin_v[k] = j < lb ? load_stream(v + j) : 0.0;
      }
    }
  }

  if (C16) check_ci_guard(ci, c16, stale);
  if (C16 && stage_fast_ok(t1, nnz)) // (the one tile that holds the ragged end of the arrays reads colindex as usual; the engine vouches for x32)
    stage_products_c16<kThreads, NPT, NTC, NTV>(lds, t0, t0, t1, c16, v, x);
  else if (EARLY && early) stage_finish<kThreads, EARLY ? NPT : 4>(lds, early_regs, x);
  else if (NTC && NTV && cache_ends > 0 && (t < cache_ends || t >= ntiles - cache_ends)) // (k_rowblock.hip: cacheable grid ends)
    stage_products<kThreads, NPT, false, false, HINT>(lds, t0, t1, nnz, ci, v, x, xcd_chunk >= 0, cold, (reverse & 2) != 0);
  else stage_products<kThreads, NPT, NTC, NTV, HINT>(lds, t0, t1, nnz, ci, v, x, xcd_chunk >= 0, cold, (reverse & 2) != 0);

  double fin_extra = 0.0; // this lane's share of the overhang (first wave, finish mode)
  if (fin) {
#pragma unroll
    for (int k = 0; k < FIN_STEPS; ++k)
      if (fin_c[k] >= 0) fin_extra += fin_v[k] * x[fin_c[k]];
  }

  __shared__ unsigned char seg_head[SEGSUM ? STRIDE : 1];
  __shared__ double seg_wave_sum[kThreads / kWave];
  __shared__ int seg_wave_flag[kThreads / kWave];
  if (SEGSUM) { // flags cleared before the barrier that publishes the tile
#pragma unroll
    for (int e = 0; e < NPT; ++e) seg_head[NPT * threadIdx.x + e] = 0;
  }

  __syncthreads();

  if (SEGSUM) {
    // 1. flag the first product of every row that starts inside the tile (one lane per row; the tile start opens a segment too)
    if (threadIdx.x == 0) seg_head[0] = 1;
    for (int base = 0; base < nrows; base += kThreads) {
      const int i = base + static_cast<int>(threadIdx.x);
      if (i < nrows) {
        const int a = (base == 0) ? a0 : rp[first + i];
        const int b = (base == 0) ? b0 : rp[first + i + 1];
        if (a >= t0 && a < t1 && b > a) seg_head[a - t0] = 1;
      }
    }
    __syncthreads();
    // 2. every lane scans its NPT consecutive products in place
    const int c0 = NPT * static_cast<int>(threadIdx.x);
    double run = 0.0;
    int first_head = NPT; // position of the first flag in this lane's chunk (NPT: none)
#pragma unroll
    for (int e = 0; e < NPT; ++e) {
      if (seg_head[c0 + e]) {
        run = 0.0;
        first_head = first_head == NPT ? e : first_head;
      }
      run += lds[c0 + e];
      lds[c0 + e] = run;
    }
    // 3. segmented inclusive scan of the lanes' open sums: (flag, value) pairs, flag = the chunk holds a row start
    double sv = run;
    int sf = first_head < NPT ? 1 : 0;
    const int wl = threadIdx.x & (kWave - 1);
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
      const double pv = __shfl_up(sv, d, kWave);
      const int pf = __shfl_up(sf, d, kWave);
      if (wl >= d && !sf) {
        sv += pv;
        sf = pf;
      }
    }
    if (wl == kWave - 1) {
      seg_wave_sum[threadIdx.x / kWave] = sv;
      seg_wave_flag[threadIdx.x / kWave] = sf;
    }
    __syncthreads();
    double incoming = 0.0; // what flows into this wave from the waves before it
    for (int k = 0; k < static_cast<int>(threadIdx.x / kWave); ++k) incoming = seg_wave_flag[k] ? seg_wave_sum[k] : incoming + seg_wave_sum[k];
    double ev = __shfl_up(sv, 1, kWave);
    int ef = __shfl_up(sf, 1, kWave);
    if (wl == 0) {
      ev = 0.0;
      ef = 0;
    }
    const double carry_in = ef ? ev : ev + incoming;
    // 4. the products before the chunk's first flag belong to the segment that was open on entry
#pragma unroll
    for (int e = 0; e < NPT; ++e)
      if (e < first_head) lds[c0 + e] += carry_in;
    __syncthreads();
  }

  // all lanes walk the same number of iterations so the DPP reduction sees a full exec mask
  for (int base = 0; base < nrows; base += vecs) {
    const int r = first + base + vec_id;
    const bool live = (base + vec_id) < nrows;
    int a = a0, b = b0;
    if (base > 0) {
      a = b = 0;
      if (live) {
        a = rp[r];
        b = rp[r + 1];
      }
    }
    const int lo = (a > t0 ? a : t0) - t0;
    const int hi = (b < t1 ? b : t1) - t0;
    double s;
    if (SEGSUM) {
      s = hi > lo ? lds[hi - 1] : 0.0; // the scanned value at the row's last product in the tile
    } else {
      s = tile_row_sum<kThreads>(lds, spans, lo, hi > lo ? hi : lo, lane, w); // long spans go to whole waves
      s = group_sum_dyn(s, w);
    }
    if (live && lane == 0) {
      if (a >= t0 && b <= t1) {
        // complete row (possibly empty): final value
        if (base == 0 && early_y) y[r] = alpha * s + beta * y_old0;
        else store_y(y, yin, r, alpha, beta, s);
      } else if (a < t0) {
        // row started in an earlier tile.  Its owner (the tile it starts in) finishes a short overhang itself; only a
        // row that runs more than kFlatFinish non-zeros past its owner's end is folded from carries.
        const long long owner_end = (static_cast<long long>(a) / STRIDE + 1) * STRIDE;
        if (b - owner_end > reach) head[t] = s;
      } else if (b - t1 <= reach) {
        sh_tail_sum = s; // row starts here and ends at most kFlatFinish non-zeros into the next tile(s): finished below
        sh_tail_row = r;
        sh_tail_yold = (base == 0 && early_y) ? y_old0 : (early_y ? yin[r] : 0.0);
      } else {
        tail[t] = s; // long row: carry, folded by the fix-up kernel in tile order
        tail_row[t] = r;
        tail_end[t] = b;
      }
    }
  }
  if (reach > 0) __syncthreads();
  // The first wave adds up the overhang it fetched at the start (<= kFlatFinish non-zeros, kFlatFinish / 64 per lane) and
  // completes y[row]: no carry, no second kernel for such rows.  (sh_tail_row >= 0 exactly when `fin` was set: only the
  // tile's last row can run past t1.)
  if (reach > 0 && threadIdx.x < kWave) {
    const int r = sh_tail_row;
    const double extra = group_sum<64>(fin_extra);
    if (r >= 0 && threadIdx.x == 0) y[r] = (beta == 0.0) ? alpha * (sh_tail_sum + extra) : alpha * (sh_tail_sum + extra) + beta * sh_tail_yold;
  }
  // a tile without a carried row says so (the fix-up reads tail_row only)
  if (threadIdx.x == 0) {
    const bool has_carry = nrows > 0 && rec.z >= t0 && rec.w - t1 > reach;
    if (!has_carry) tail_row[t] = -1;
  }
}

// Plan time, one thread per tile: the tile's digest {first row, end row (exclusive), rowptr[end - 1], rowptr[end]} (zeros for
// the extents of a tile that owns no rows).
__global__ __launch_bounds__(256) void flat_digest_kernel(const int *__restrict__ rp, const int *__restrict__ bp, int ntiles, int m,
                                                          int nnz, int stride, int tile0, int4v *__restrict__ dig) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  const int t0 = (t + tile0) * stride;
  const int t1 = (nnz - t0 > stride) ? t0 + stride : nnz;
  int first = bp[t];
  first = first < m ? first : m;
  const int end_excl = tile_end_excl(rp, bp, t, ntiles, m, t1);
  int4v rec;
  rec.x = first;
  rec.y = end_excl;
  rec.z = end_excl > first ? rp[end_excl - 1] : 0;
  rec.w = end_excl > first ? rp[end_excl] : 0;
  dig[t] = rec;
}

// Plan time, one thread per tile:  flag[0] = 1 if any row runs more than kFlatFinish non-zeros past the end of the tile it
// starts in (only then does the fix-up kernel have anything to fold);  flag[1] = the largest number of rows any tile owns
// (a tile is one workgroup: tens of thousands of -- mostly empty -- rows in one tile serialise there).
__global__ __launch_bounds__(256) void flat_needs_fixup_kernel(const int *__restrict__ rp, const int *__restrict__ bp,
                                                               int ntiles, int m, int nnz, int stride, int tile0,
                                                               int *__restrict__ flag) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ntiles) return;
  const int t0 = (t + tile0) * stride;
  const int t1 = (nnz - t0 > stride) ? t0 + stride : nnz;
  int first = bp[t];
  first = first < m ? first : m;
  const int end_excl = tile_end_excl(rp, bp, t, ntiles, m, t1);
  if (end_excl <= first) return;
  atomicMax(flag + 1, end_excl - first);
  if (t == ntiles - 1) return; // the last tile cannot have a row that continues
  const int r = end_excl - 1;
  if (rp[r] >= t0 && rp[r + 1] - t1 > kFlatFinish) *flag = 1; // idempotent store
}

// One thread per tile that holds the START of a cut row: adds its tail carry and the head carries of
// the following tiles in tile order, then applies alpha/beta once.  Everything it needs was written by
// the tile kernel, so the dependent-load chain is one level deep.
__global__ __launch_bounds__(256) void flat_fixup_kernel(int ntiles, int stride, int tile0, double alpha, double beta,
                                                         const double *__restrict__ head,
                                                         const double *__restrict__ tail,
                                                         const int *__restrict__ tail_row,
                                                         const int *__restrict__ tail_end, double *y, const double *yin) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  // no early return: every lane of a wave takes part in wave_range_sum
  int r = -1;
  if (t < ntiles - 1) r = tail_row[t]; // the last tile cannot have a row that continues
  int k0 = 0, k1 = 0;
  double s = 0.0;
  if (r >= 0) {
    const long long b = tail_end[t];
    k0 = t + 1;                                            // heads of the tiles the row runs through:
    k1 = static_cast<int>((b + stride - 1) / stride) - tile0; // ... up to the tile that holds its last non-zero
    k1 = k1 < ntiles ? k1 : ntiles;
    s = tail[t];
  }
  s += wave_range_sum(head, k0, k1);
  if (r >= 0) store_y(y, yin, r, alpha, beta, s);
}

} // namespace

void launch_break_points_from(hipStream_t stream, const int *rp, int m, int nnz, int stride, int tile0, int *bp, int bp_len) {
  if (bp_len <= 0) return;
  SPMV_ACC_LAUNCH(break_points_kernel, dim3((bp_len + 255) / 256), dim3(256), 0, stream, rp, m, nnz, stride, tile0, bp,
                     bp_len);
}
void launch_break_points(hipStream_t stream, const int *rp, int m, int nnz, int stride, int *bp, int bp_len) {
  launch_break_points_from(stream, rp, m, nnz, stride, 0, bp, bp_len);
}

namespace {
template <int NPT, bool NTC, bool NTV>
void launch_flat_variant(hipStream_t stream, const CsrDev &A, const FlatPlan &P, double alpha, double beta, const double *x,
                         double *y) {
  const Col16Dev none = {};
  if (P.early_stream)
    SPMV_ACC_LAUNCH((flat_tile_kernel<NPT, NTC, NTV, true>), dim3(P.ntiles), dim3(kThreads), 0, stream, A.m, A.nnz,
                       P.ntiles, alpha, beta, A.rp, P.bp, A.ci, A.v, x, y, A.yin ? A.yin : y, P.head, P.tail, P.tail_row, P.tail_end,
                       P.xcd_chunk, P.needs_fixup ? 0 : kFlatFinish, P.tile0, A.guard, A.stale, none, (P.reverse ? 1 : 0) | (x32_ok(A) ? 2 : 0), P.cache_ends, static_cast<const int4v *>(P.digest), nullptr);
  else
    SPMV_ACC_LAUNCH((flat_tile_kernel<NPT, NTC, NTV, false>), dim3(P.ntiles), dim3(kThreads), 0, stream, A.m, A.nnz,
                       P.ntiles, alpha, beta, A.rp, P.bp, A.ci, A.v, x, y, A.yin ? A.yin : y, P.head, P.tail, P.tail_row, P.tail_end,
                       P.xcd_chunk, P.needs_fixup ? 0 : kFlatFinish, P.tile0, A.guard, A.stale, none, (P.reverse ? 1 : 0) | (x32_ok(A) ? 2 : 0), P.cache_ends, static_cast<const int4v *>(P.digest), nullptr);
}
// the segmented-scan reduction (reference option FLAT_SEGMENT_SUM_REDUCE): 2048-non-zero tiles, values / colindex under the plan's policy
template <bool NTC, bool NTV>
void launch_flat_segsum(hipStream_t stream, const CsrDev &A, const FlatPlan &P, double alpha, double beta, const double *x,
                        double *y) {
  const Col16Dev none = {};
  SPMV_ACC_LAUNCH((flat_tile_kernel<kNnzPerThread, NTC, NTV, false, false, true>), dim3(P.ntiles), dim3(kThreads), 0, stream, A.m,
                     A.nnz, P.ntiles, alpha, beta, A.rp, P.bp, A.ci, A.v, x, y, A.yin ? A.yin : y, P.head, P.tail, P.tail_row, P.tail_end,
                     P.xcd_chunk, P.needs_fixup ? 0 : kFlatFinish, P.tile0, A.guard, A.stale, none, (P.reverse ? 1 : 0) | (x32_ok(A) ? 2 : 0), P.cache_ends,
                     static_cast<const int4v *>(P.digest), nullptr);
}
// gather hints: 2048-non-zero tiles, break-point chain first
template <bool NTC, bool NTV>
void launch_flat_hint(hipStream_t stream, const CsrDev &A, const FlatPlan &P, double alpha, double beta, const double *x, double *y) {
  const Col16Dev none = {};
  SPMV_ACC_LAUNCH((flat_tile_kernel<kNnzPerThread, NTC, NTV, false, false, false, true>), dim3(P.ntiles), dim3(kThreads), 0, stream, A.m,
                     A.nnz, P.ntiles, alpha, beta, A.rp, P.bp, A.ci, A.v, x, y, A.yin ? A.yin : y, P.head, P.tail, P.tail_row, P.tail_end,
                     P.xcd_chunk, P.needs_fixup ? 0 : kFlatFinish, P.tile0, A.guard, A.stale, none, (P.reverse ? 1 : 0) | (x32_ok(A) ? 2 : 0), P.cache_ends,
                     static_cast<const int4v *>(P.digest), A.cold);
}
// 16-bit columns (the plan's encoding, timed per matrix): NPT 8 tiles (a multiple of the 256-non-zero chunk), offsets / values under the plan's cache policy
template <bool NTC, bool NTV>
void launch_flat_col16(hipStream_t stream, const CsrDev &A, const FlatPlan &P, double alpha, double beta, const double *x,
                       double *y) {
  const Col16Dev c = col16_dev(*P.col16, A);
  SPMV_ACC_LAUNCH((flat_tile_kernel<kNnzPerThread, NTC, NTV, false, true>), dim3(P.ntiles), dim3(kThreads), 0, stream, A.m,
                     A.nnz, P.ntiles, alpha, beta, A.rp, P.bp, A.ci, A.v, x, y, A.yin ? A.yin : y, P.head, P.tail, P.tail_row, P.tail_end,
                     P.xcd_chunk, P.needs_fixup ? 0 : kFlatFinish, P.tile0, A.guard, A.stale, c, (P.reverse ? 1 : 0) | 2, P.cache_ends, static_cast<const int4v *>(P.digest), nullptr);
}
template <int NPT>
void launch_flat_policy(hipStream_t stream, const CsrDev &A, const FlatPlan &P, double alpha, double beta, const double *x,
                        double *y) {
  switch (P.stream_policy & 3) { // cache policy of the stream loads, see kernels.hpp
  case 1: launch_flat_variant<NPT, false, false>(stream, A, P, alpha, beta, x, y); break;
  case 2: launch_flat_variant<NPT, false, true>(stream, A, P, alpha, beta, x, y); break;
  case 3: launch_flat_variant<NPT, true, false>(stream, A, P, alpha, beta, x, y); break;
  default: launch_flat_variant<NPT, true, true>(stream, A, P, alpha, beta, x, y); break;
  }
}
} // namespace

void launch_flat_digest(hipStream_t stream, const CsrDev &A, const FlatPlan &P) {
  if (P.ntiles <= 0) return;
  SPMV_ACC_LAUNCH(flat_digest_kernel, dim3((P.ntiles + 255) / 256), dim3(256), 0, stream, A.rp, P.bp, P.ntiles, A.m, A.nnz, P.stride,
                     P.tile0, static_cast<int4v *>(P.digest));
}

void launch_flat_needs_fixup(hipStream_t stream, const CsrDev &A, const FlatPlan &P, int *d_flag) {
  if (P.ntiles <= 0) return;
  SPMV_ACC_LAUNCH(flat_needs_fixup_kernel, dim3((P.ntiles + 255) / 256), dim3(256), 0, stream, A.rp, P.bp, P.ntiles,
                     A.m, A.nnz, P.stride, P.tile0, d_flag);
}

void launch_flat(hipStream_t stream, const CsrDev &A, const FlatPlan &P, double alpha, double beta, const double *x,
                 double *y) {
  if (P.ntiles <= 0) return;
  const int npt = P.stride / kThreads;
  if (P.segment_sum && npt == kNnzPerThread) {
    switch (P.stream_policy & 3) {
    case 1: launch_flat_segsum<false, false>(stream, A, P, alpha, beta, x, y); break;
    case 2: launch_flat_segsum<false, true>(stream, A, P, alpha, beta, x, y); break;
    case 3: launch_flat_segsum<true, false>(stream, A, P, alpha, beta, x, y); break;
    default: launch_flat_segsum<true, true>(stream, A, P, alpha, beta, x, y); break;
    }
  } else if (A.cold != nullptr && !P.col16 && !P.early_stream && npt == kNnzPerThread) {
    switch (P.stream_policy & 3) {
    case 1: launch_flat_hint<false, false>(stream, A, P, alpha, beta, x, y); break;
    case 2: launch_flat_hint<false, true>(stream, A, P, alpha, beta, x, y); break;
    case 3: launch_flat_hint<true, false>(stream, A, P, alpha, beta, x, y); break;
    default: launch_flat_hint<true, true>(stream, A, P, alpha, beta, x, y); break;
    }
  } else if (P.col16 && P.col16->state == 1 && x32_ok(A) && npt == kNnzPerThread) {
    switch (P.stream_policy & 3) {
    case 1: launch_flat_col16<false, false>(stream, A, P, alpha, beta, x, y); break;
    case 2: launch_flat_col16<false, true>(stream, A, P, alpha, beta, x, y); break;
    case 3: launch_flat_col16<true, false>(stream, A, P, alpha, beta, x, y); break;
    default: launch_flat_col16<true, true>(stream, A, P, alpha, beta, x, y); break;
    }
  } else if (npt == 4) launch_flat_policy<4>(stream, A, P, alpha, beta, x, y);
  else if (npt == 16) launch_flat_policy<16>(stream, A, P, alpha, beta, x, y);
  else launch_flat_policy<8>(stream, A, P, alpha, beta, x, y);
  if (P.ntiles > 1 && P.needs_fixup) {
    SPMV_ACC_LAUNCH(flat_fixup_kernel, dim3((P.ntiles - 1 + 255) / 256), dim3(256), 0, stream, P.ntiles, P.stride,
                       P.tile0, alpha, beta, P.head, P.tail, P.tail_row, P.tail_end, y, A.yin ? A.yin : y);
  }
}

} // namespace spmv_acc
