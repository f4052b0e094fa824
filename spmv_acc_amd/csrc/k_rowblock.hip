// k_rowblock.hip -- LINE_ENHANCE family: fixed row blocks, non-zeros streamed through LDS in rounds.
//
// Reference roles (all reached from KERNEL_STRATEGY=LINE_ENHANCE / LINE / ADAPTIVE):
//   * hip-line-enhance/line_enhance_spmv_imp.inl:12-95 (line_enhance_kernel) with the parameter
//     picks of line_enhance_spmv.cpp:8-69 (line_enhance_sparse_spmv, adaptive_enhance_sparse_spmv)
//   * hip-line/line_adaptive_one_pass.inl:42-115 (spmv_adaptive_line_kernel) -- the short-row case
//     is the VEC=1 instance of the same structure.
// What is kept: a workgroup owns consecutive rows, their non-zeros are multiplied into an LDS tile
// cooperatively (coalesced, independent of row boundaries), each row's VEC lanes then add the part
// of the row that lies in the tile, and partial sums live in registers across rounds.
// What is different (MI355X-first):
//   * y = alpha*s + beta*y for any beta (the reference stores alpha*s + y, i.e. beta = 1 only)
//   * rows per workgroup = THREADS/VEC so every lane has a row to reduce (the reference leaves
//     3/4 of the lanes idle in the reduce phase for RPB=32, VEC=4, THREADS=512)
//   * 16-B non-temporal loads of colindex/value, 4 non-zeros per lane per step, tile start aligned
//     down to a multiple of 4 so the wide loads stay aligned for any row-block start
//   * optional XCD-contiguous block order: neighbouring row blocks (which share x[] lines for
//     banded / FEM matrices) run on the same XCD and hit in its 4 MB L2
#include "device_utils.hpp"
#include "kernels.hpp"
#include "tile_stage.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

// Row extents from the plan's row digest instead of rowptr (LENS): one byte per row (its length) and one int per row block
// (the block's first non-zero; bit 31 set = some row of the block is longer than 255, the block then reads rowptr as usual).
// A row's extent is the block's base plus a scan of the lengths: the lanes of a row all hold its length, the row's leader lane
// feeds it into a wave scan (DPP row_shr + three lane reads), and the four waves' totals cross through LDS behind the barrier
// that follows staging -- no barrier is added.  4 B/row of rowptr traffic become 1 B/row + 4 B/block: on the Hardesty3-sized
// matrix (4.9 nnz/row) that is 24.6 MB of 710 MB per SpMV.
template <int CTRL> __device__ __forceinline__ int dpp_shr_or_zero(int v) {
  return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, false); // lanes without a source lane get 0
}
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += dpp_shr_or_zero<0x111>(v); // row_shr:1
  v += dpp_shr_or_zero<0x112>(v); // row_shr:2
  v += dpp_shr_or_zero<0x114>(v); // row_shr:4
  v += dpp_shr_or_zero<0x118>(v); // row_shr:8 -> inclusive scan inside each row of 16 lanes
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
  const int r = (threadIdx.x & (kWave - 1)) >> 4;
  return v + (r > 0 ? t0 : 0) + (r > 1 ? t1 : 0) + (r > 2 ? t2 : 0);
}

// HINT: gather hints (k_hint.hip): the block's gathers take their cache policy from the plan's cold bits.
// C16: the non-zeros' columns come from the plan's 16-bit encoding (k_col16.hip) instead of colindex: 2.25 B instead of 4 B of stream per
// non-zero.  The tile origin is then aligned down to a 256-non-zero chunk (a wavefront's step = one chunk record) instead of to 4.
template <int VEC, bool NTC, bool NTV, bool LENS, bool HINT = false, bool C16 = false>
__global__ __launch_bounds__(kThreads) void rowblock_stream_kernel(int m, int nnz, int nblocks, int rpb, int flags,
                                                                   double alpha, double beta,
                                                                   const int *__restrict__ rp,
                                                                   const int *__restrict__ ci,
                                                                   const double *__restrict__ v,
                                                                   const double *__restrict__ x,
                                                                   double *y, const double *yin,
                                                                   const int *__restrict__ guard,
                                                                   int *__restrict__ stale,
                                                                   const unsigned char *__restrict__ lens,
                                                                   const int *__restrict__ base, int cache_ends,
                                                                   const unsigned char *__restrict__ cold, Col16Dev c16) {
  check_plan_guard(rp, m, guard, stale);
  if (C16) check_ci_guard(ci, c16, stale);
  __shared__ int wave_tot[kThreads / kWave];
  // rpb rows per workgroup, rpb <= kThreads / VEC (not necessarily a power of two: it is chosen so that
  // rpb * average row length fills most of one LDS tile)
  __shared__ __attribute__((aligned(16))) double lds[kTile]; // written 16 B at a time
  __shared__ TileSpans2K spans;
  if (threadIdx.x == 0) spans.n = 0; // published by the barrier that follows the first staging

  int b = blockIdx.x;
  if (flags & 64) b = zigzag_block(b, nblocks); // every other SpMV on a plan walks the matrix backwards (dispatch.cpp)
  if (flags & 4) b = xcd_chunked_block(b, nblocks, flags >> 8);
  // (Tried in round 5 and removed: workgroups walking GROUPS of 2 ... 8 consecutive row blocks, so that the stores of block i travel while block
  // i + 1 streams and s_endpgm's wait for outstanding stores is paid once per group: 2-14 % SLOWER on every stand-in, small and large --
  // profiles/r05_short_row_dissection.txt.  What the y stores cost is the memory system's price for a write stream beside a read stream, not a wait.)

  const long long base_ll = static_cast<long long>(b) * rpb;
  const int row_base = static_cast<int>(base_ll);
  const int row_end = (base_ll + rpb < m) ? row_base + rpb : m;
  // wave-uniform: the non-zero range of the whole block
  int s0, s1;
  bool from_lens = false;
  if (LENS) {
    const int b0 = base[b];
    s0 = b0 & 0x7fffffff;
    s1 = base[b + 1] & 0x7fffffff;
    from_lens = b0 >= 0;
  } else {
    s0 = rp[row_base];
    s1 = rp[row_end];
  }

  const int lane = threadIdx.x % VEC;
  const int row = row_base + threadIdx.x / VEC;
  const bool live = row < row_end; // lanes beyond rpb * VEC only help staging
  int r0 = 0, r1 = 0, len = 0;
  if (LENS && from_lens) {
    if (live) len = lens[row];
  } else if (live) {
    r0 = rp[row];
    r1 = rp[row + 1];
  }
  // the old y is needed only at the very end: ask for it now so its latency hides behind the whole tile
  const bool writer = live && lane == 0;
  double y_old = 0.0;
  if (beta != 0.0 && writer) y_old = yin[row]; // (asked for now, used at the very end; loading it late was an A/B switch until round 5: never faster)

  double acc = 0.0;
  int incl = 0;
  // tile origin aligned down so 16-B loads stay aligned; the (at most 3) extra leading products are never read (C16: aligned down to the
  // chunk, the lanes in front of the block's first group re-read that group)
  const int lo4 = s0 & ~3;
  // (an EMPTY block whose first group starts at s1 stages nothing: lo4 >= s1 ends the loop before it starts, as without the encoding)
  const int off0 = (C16 && lo4 < s1) ? (s0 & ~(kCol16Chunk - 1)) : lo4;
  const bool c16_ok = C16 && stage_fast_ok(s1, nnz); // (the one block that holds the ragged end of the arrays reads colindex as usual)
  for (int off = off0; off < s1; off += kTile) {
    // Plans that stream non-temporally (short rows: the vectors are worth more cache than the matrix) still keep the two ENDS
    // of the grid cacheable -- `cache_ends` blocks each, ~24 MB of stream, an L2's worth: with the zigzag order those are the
    // blocks the next SpMV starts with (Hardesty3-sized 155.0 -> 152.9 us; 8 / 16 / 24 / 32 / 48 / 100 MB: 153.5 / 152.9 / 152.9 /
    // 152.9 / 153.3 / 155.1).
    const bool cached_end = NTC && NTV && cache_ends > 0 && (b < cache_ends || b >= nblocks - cache_ends);
    if (C16 && c16_ok) {
      if (cached_end) stage_products_c16<kThreads, kNnzPerThread, false, false>(lds, off, lo4 > off ? lo4 : off, s1, c16, v, x);
      else stage_products_c16<kThreads, kNnzPerThread, NTC, NTV>(lds, off, lo4 > off ? lo4 : off, s1, c16, v, x);
    } else if (cached_end)
      stage_products<kThreads, kNnzPerThread, false, false, HINT>(lds, off, s1, nnz, ci, v, x, true, cold, (flags & 128) != 0);
    else
      stage_products<kThreads, kNnzPerThread, NTC, NTV, HINT>(lds, off, s1, nnz, ci, v, x, true, cold, (flags & 128) != 0);
    if (LENS && from_lens && off == off0) { // (block-uniform branch; the scan needs every lane of the wave)
      // The row lengths are consumed HERE, behind the staging: scanned in front of it (rounds 2-5) the scan's wait for lens[row] stood between
      // the block's bounds and its first stream load -- one more dependent round trip per workgroup, for a byte the tile does not need.
      incl = wave_inclusive_scan(lane == 0 ? len : 0);
      if ((threadIdx.x & (kWave - 1)) == kWave - 1) wave_tot[threadIdx.x / kWave] = incl;
    }
    __syncthreads();
    if (LENS && from_lens && off == off0) {
      const int w = threadIdx.x / kWave;
      int before = s0;
#pragma unroll
      for (int k = 0; k < kThreads / kWave - 1; ++k) before += (k < w) ? wave_tot[k] : 0;
      r1 = before + incl; // lanes of one row hold the same inclusive sum: the row's leader is the row's first lane
      r0 = r1 - len;
    }
    const int lo = (r0 > off ? r0 : off) - off;
    const int hi = (r1 < off + kTile ? r1 : off + kTile) - off;
    acc += tile_row_sum<kThreads>(lds, spans, lo, hi > lo ? hi : lo, lane, VEC); // long spans go to whole waves
    if (off + kTile < s1) __syncthreads(); // next round overwrites the tile
  }
  acc = group_sum<VEC>(acc);
  if (writer) {
    // non-temporal store (round 5): a write stream beside a read stream costs the memory system ~3 x its bytes (read_write_mix_bench.hip: 24 KB read +
    // 2 KB written per workgroup reads at 5.1-5.7 TB/s, 5.4-5.9 with non-temporal stores); in this kernel: banded shard -2.1 %, the other stand-ins
    // -0.1 ... -0.6 %, none slower (two builds A/B-ed in one process, profiles/r05_short_row_dissection.txt)
    __builtin_nontemporal_store(beta == 0.0 ? alpha * acc : alpha * acc + beta * y_old, y + row);
  }
}

// Plan-time balance probe over the workgroups of `rpb` consecutive rows the row-block kernel would use:
//   out[0] = largest number of non-zeros any of them would own,
//   out[1] = how many of them are more than 35 % away from the average block (`avg_block` non-zeros) in either direction, or
//            spill over one 2048-product tile.
__global__ __launch_bounds__(256) void max_block_nnz_kernel(const int *__restrict__ rp, int m, int rpb, int nblocks,
                                                            int avg_block, int *__restrict__ out) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  int v = 0;
  bool off = false;
  if (b < nblocks) {
    const long long lo = static_cast<long long>(b) * rpb;
    const long long hi = lo + rpb < m ? lo + rpb : m;
    v = rp[hi] - rp[lo];
    const long long d = static_cast<long long>(v) - avg_block;
    // the ragged last block does not count; a block that needs a second (mostly empty) LDS round counts whatever its distance
    off = hi - lo == rpb && (20 * (d < 0 ? -d : d) > 7LL * avg_block || v > kTile);
  }
  const unsigned long long votes = __ballot(off);
  // wave max, then one atomic per wave
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const int other = __shfl_xor(v, o, 64);
    v = other > v ? other : v;
  }
  if ((threadIdx.x & 63) == 0) {
    if (v > 0) atomicMax(out, v);
    if (votes) atomicAdd(out + 1, __popcll(votes));
  }
}

// Plan time: lens[r] = min(rowptr[r+1] - rowptr[r], 255); base[b] = rowptr[b * rpb], bit 31 set when a row of block b is
// longer than 255 (base is initialised by the first kernel, flagged by the second).
__global__ __launch_bounds__(256) void row_digest_base_kernel(const int *__restrict__ rp, int m, int rpb, int nblocks,
                                                              int *__restrict__ base) {
  const int b = blockIdx.x * 256 + threadIdx.x;
  if (b > nblocks) return;
  const long long r = static_cast<long long>(b) * rpb;
  base[b] = rp[r < m ? r : m];
}
__global__ __launch_bounds__(256) void row_digest_lens_kernel(const int *__restrict__ rp, int m, int rpb,
                                                              unsigned char *__restrict__ lens, int *__restrict__ base) {
  const long long r = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (r >= m) return;
  const int len = rp[r + 1] - rp[r];
  lens[r] = static_cast<unsigned char>(len < 255 ? (len > 0 ? len : 0) : 255);
  if (len > 255 || len < 0) atomicOr(base + r / rpb, static_cast<int>(0x80000000u));
}

template <int VEC, bool NC, bool NV, bool LN, bool H, bool C>
void launch_rb_instance(hipStream_t stream, const CsrDev &A, const RowDigest *D, const Col16Dev &c16, int nblocks, int rpb, int flags,
                        double alpha, double beta, const double *x, double *y, int cache_ends) {
  SPMV_ACC_LAUNCH((rowblock_stream_kernel<VEC, NC, NV, LN, H, C>), dim3(nblocks), dim3(kThreads), 0, stream, A.m, A.nnz, nblocks, rpb, flags,
                     alpha, beta, A.rp, A.ci, A.v, x, y, A.yin ? A.yin : y, A.guard, A.stale,
                     LN ? D->lens : static_cast<const unsigned char *>(nullptr), LN ? D->base : static_cast<const int *>(nullptr), cache_ends,
                     H ? A.cold : static_cast<const unsigned char *>(nullptr), c16);
}

// Instances: the row digest (LENS) exists for lane groups of 1 and 2 only -- it is used for rows of <= 8 non-zeros on average, where the shape
// rule gives every row one lane; a digest asked for with wider groups is ignored (the kernel reads rowptr: same result).  Gather hints and the
// 16-bit columns exclude each other (hints serve power-law columns, the encoding local ones): three staging forms per policy.
template <int VEC, bool NC, bool NV>
void launch_policy(hipStream_t stream, const CsrDev &A, const RowDigest *D, const Col16 *C, int nblocks, int rpb, int flags, double alpha,
                   double beta, const double *x, double *y, int cache_ends) {
  const bool lens = VEC <= 2 && D && D->lens;
  const bool c16 = C && C->state == 1 && A.cold == nullptr && x32_ok(A);
  const Col16Dev cd = c16 ? col16_dev(*C, A) : Col16Dev{};
#define SPMV_ACC_RB(LN, H, CC) launch_rb_instance<VEC, NC, NV, LN, H, CC>(stream, A, D, cd, nblocks, rpb, flags, alpha, beta, x, y, cache_ends)
  if constexpr (VEC <= 2) {
    if (lens) {
      if (c16) SPMV_ACC_RB(true, false, true);
      else if (A.cold) SPMV_ACC_RB(true, true, false);
      else SPMV_ACC_RB(true, false, false);
      return;
    }
  }
  if (c16) SPMV_ACC_RB(false, false, true);
  else if (A.cold) SPMV_ACC_RB(false, true, false);
  else SPMV_ACC_RB(false, false, false);
#undef SPMV_ACC_RB
}

template <int VEC>
void launch_vec(hipStream_t stream, const CsrDev &A, const RowDigest *D, const Col16 *C, int rpb, int xcd, double alpha, double beta,
                const double *x, double *y, int cache_ends) {
  if (rpb < 1 || rpb > kThreads / VEC) rpb = kThreads / VEC;
  if (D && D->rpb != rpb) D = nullptr; // a digest built for another block size does not apply
  const int nblocks = static_cast<int>((static_cast<long long>(A.m) + rpb - 1) / rpb);
  if (nblocks == 0) return;
  const int remap = (xcd & ~0xb) | (x32_ok(A) ? 128 : 0); // bit 2 + bits 8..: XCD-chunked order, bits 4-5: stream policy, bit 6: zigzag, bit 7: 32-bit gather offsets
  // bits 4-5 of the flags: cache policy of the stream loads (0 nt/nt, 1 plain/plain, 2 colindex plain + values nt,
  // 3 colindex nt + values plain).  One set of kernels for every base-pointer alignment: their 16-B loads go through under-aligned
  // vector types (device_utils.hpp), the same instruction with the same cache policy whether or not the caller's arrays are 16-B aligned
  switch ((xcd >> 4) & 3) {
  case 1: launch_policy<VEC, false, false>(stream, A, D, C, nblocks, rpb, remap, alpha, beta, x, y, cache_ends); break;
  case 2: launch_policy<VEC, false, true>(stream, A, D, C, nblocks, rpb, remap, alpha, beta, x, y, cache_ends); break;
  case 3: launch_policy<VEC, true, false>(stream, A, D, C, nblocks, rpb, remap, alpha, beta, x, y, cache_ends); break;
  default: launch_policy<VEC, true, true>(stream, A, D, C, nblocks, rpb, remap, alpha, beta, x, y, cache_ends); break;
  }
}

} // namespace

void launch_max_block_nnz(hipStream_t stream, const int *rp, int m, int rows_per_block, int avg_block, int *d_out) {
  const int nblocks = static_cast<int>((static_cast<long long>(m) + rows_per_block - 1) / rows_per_block);
  if (nblocks <= 0) return;
  SPMV_ACC_LAUNCH(max_block_nnz_kernel, dim3((nblocks + 255) / 256), dim3(256), 0, stream, rp, m, rows_per_block,
                     nblocks, avg_block, d_out);
}

void launch_row_digest(hipStream_t stream, const int *rp, int m, int rows_per_block, unsigned char *lens, int *base) {
  if (m <= 0 || rows_per_block <= 0) return;
  const int nblocks = static_cast<int>((static_cast<long long>(m) + rows_per_block - 1) / rows_per_block);
  SPMV_ACC_LAUNCH(row_digest_base_kernel, dim3((nblocks + 1 + 255) / 256), dim3(256), 0, stream, rp, m, rows_per_block, nblocks,
                     base);
  SPMV_ACC_LAUNCH(row_digest_lens_kernel, dim3(static_cast<unsigned>((static_cast<long long>(m) + 255) / 256)), dim3(256), 0,
                     stream, rp, m, rows_per_block, lens, base);
}

void launch_rowblock_stream(hipStream_t stream, const CsrDev &A, int vec, int rows_per_block, int xcd_remap,
                            double alpha, double beta, const double *x, double *y, const RowDigest *digest, int cache_ends,
                            const Col16 *col16) {
  switch (vec) {
  case 1: launch_vec<1>(stream, A, digest, col16, rows_per_block, xcd_remap, alpha, beta, x, y, cache_ends); break;
  case 2: launch_vec<2>(stream, A, digest, col16, rows_per_block, xcd_remap, alpha, beta, x, y, cache_ends); break;
  case 4: launch_vec<4>(stream, A, digest, col16, rows_per_block, xcd_remap, alpha, beta, x, y, cache_ends); break;
  case 8: launch_vec<8>(stream, A, digest, col16, rows_per_block, xcd_remap, alpha, beta, x, y, cache_ends); break;
  case 16: launch_vec<16>(stream, A, digest, col16, rows_per_block, xcd_remap, alpha, beta, x, y, cache_ends); break;
  case 32: launch_vec<32>(stream, A, digest, col16, rows_per_block, xcd_remap, alpha, beta, x, y, cache_ends); break;
  default: launch_vec<64>(stream, A, digest, col16, rows_per_block, xcd_remap, alpha, beta, x, y, cache_ends); break;
  }
}

// rows per workgroup so that rows * (nnz/m) ~ target products, and the widest lane group that still gives
// every row of the workgroup its own vector
void pick_rowblock_shape(int m, int nnz, int target, int *vec, int *rows_per_block) {
  const double avg = m > 0 ? static_cast<double>(nnz) / m : 0.0;
  long long rpb = avg > 0.0 ? static_cast<long long>(target / avg) : kThreads;
  if (rpb < 1) rpb = 1;
  if (rpb > kThreads) rpb = kThreads;
  int v = 1;
  while (v < 64 && static_cast<long long>(v) * 2 * rpb <= kThreads) v <<= 1;
  *vec = v;
  *rows_per_block = static_cast<int>(rpb);
}

} // namespace spmv_acc
