// k_col16.hip -- plan-time build of the 16-bit column encoding (round 6 form; kernels.hpp Col16, tile_stage.hpp stage_products_c16).
//
// No reference counterpart: the reference streams 4-byte colindex entries (hip-flat/flat_imp_one_pass.hpp:35-39 and
// hip-line-enhance/line_enhance_spmv_imp.inl:55-62 read 12 B per non-zero: 8 value + 4 column).  On MI355X the tile kernels move their
// bytes at the fabric's rate, so the lever left is moving fewer of them.  Where columns are local -- FEM / banded matrices: the
// non-zeros of a few neighbouring rows lie within a few hundred columns of each other -- a column is a 16-bit offset from a per-chunk
// base:
//     chunk        = 256 consecutive non-zeros, aligned in the ABSOLUTE non-zero index (one wavefront's step: 64 lanes x 4)
//     rec[c][0]    = base  = median over the 64 lanes of each lane's second-smallest column, - 32767, clamped at 0  (robust against far columns)
//     d16[j]       = colindex[j] - base                 if that fits in [0, 65534]
//                  = 0xFFFF (escape)                    otherwise
//     rec[c][1]    = number of escapes of the chunk;  rec[c][4 .. R) = its first E = R - 4 escaped columns, in non-zero order
//     rec[c][2]    = where the chunk's escapes beyond E start in the overflow list `ovf`
// R (16, 32 or 64 ints per chunk) is chosen per matrix so that at most 1 % of the chunks overflow.  Round 2's form kept base and the
// escape offsets in two arrays and all escapes in one list: the kernel then needed base[c] / esc_start[c] BEFORE it could ask for its
// escapes -- a third dependent round trip in front of the gathers -- and moved 12 % fewer bytes at 9 % less rate (profiles/
// r06_col16_counters.md).  With a fixed-stride record the record's address depends on nothing but the chunk index: it is requested
// first, together with the stream, and only a chunk with more than E escapes pays a dependent load.
// 4 B/nnz of column stream become 2 B + 4 R / 256 B (R = 16: 2.25 B/nnz).
#include <climits>

#include <rocprim/device/device_scan.hpp>

#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

constexpr int kChunk = kCol16Chunk;

// One wavefront per chunk.  Pass 1: base + escape count; stats[0] += escapes, stats[1 .. 3] += chunks with more than 12 / 28 / 60 of them.
__global__ __launch_bounds__(kThreads) void col16_base_kernel(const int *__restrict__ ci, int nnz, int chunk0, int nchunks,
                                                              int *__restrict__ base, int *__restrict__ esc_count,
                                                              unsigned long long *__restrict__ stats) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long c_ll = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave;
  if (c_ll >= nchunks) return; // wave-uniform
  const int c = static_cast<int>(c_ll);
  const long long j0 = (static_cast<long long>(chunk0) + c) * kChunk + 4 * lane;
  int col[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) col[e] = (j0 + e < nnz) ? ci[j0 + e] : INT_MAX;
  // this lane's representative: the SECOND smallest of its four columns (the smallest where it holds fewer than two).  (Round 2 took the
  // minimum: at 15 % far columns the minimum of four is a far column in up to half of the lanes of a chunk late in the matrix, the median of
  // the lanes followed them and every near column of such chunks escaped -- tests/test_gpu_col16.py::test_record_size_follows_the_escape_statistics.)
  int lo0 = col[0] < col[1] ? col[0] : col[1], hi0 = col[0] < col[1] ? col[1] : col[0];
  int lo1 = col[2] < col[3] ? col[2] : col[3], hi1 = col[2] < col[3] ? col[3] : col[2];
  const int smallest = lo0 < lo1 ? lo0 : lo1;
  int second = lo0 < lo1 ? (hi0 < lo1 ? hi0 : lo1) : (hi1 < lo0 ? hi1 : lo0);
  const int mine = second != INT_MAX ? second : smallest;
  // rank of this lane's representative among the 64 (ties broken by lane id): the lane of rank (valid - 1) / 2 holds the median
  int rank = 0;
  for (int l = 0; l < kWave; ++l) {
    const int other = __shfl(mine, l, kWave);
    rank += (other < mine || (other == mine && l < lane)) ? 1 : 0;
  }
  // lanes past the end of the arrays hold INT_MAX and rank last; the median of the valid lanes is what matters
  const int valid = __popcll(__ballot(mine != INT_MAX));
  const int want = valid > 0 ? (valid - 1) / 2 : 0;
  const unsigned long long holder = __ballot(rank == want);
  const int src = holder ? __ffsll(static_cast<long long>(holder)) - 1 : 0;
  const int median = __shfl(mine, src, kWave);
  long long b = static_cast<long long>(median) - 32767;
  if (valid == 0 || b < 0) b = 0;
  const int bs = static_cast<int>(b);
  int esc = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    if (col[e] != INT_MAX) {
      const long long d = static_cast<long long>(col[e]) - bs;
      esc += (d < 0 || d > 65534) ? 1 : 0;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) esc += __shfl_xor(esc, o, kWave);
  if (lane == 0) {
    base[c] = bs;
    esc_count[c] = esc;
    if (esc > 0) atomicAdd(stats, static_cast<unsigned long long>(esc));
    if (esc > 12) atomicAdd(stats + 1, 1ull);
    if (esc > 28) atomicAdd(stats + 2, 1ull);
    if (esc > 60) atomicAdd(stats + 3, 1ull);
  }
}

// esc_count[c] -> max(0, esc_count[c] - E), in place (the input of the exclusive scan that places the overflow escapes)
__global__ __launch_bounds__(kThreads) void col16_overflow_kernel(int *__restrict__ cnt, int nchunks, int E) {
  const long long c = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (c < nchunks) {
    const int over = cnt[c] - E;
    cnt[c] = over > 0 ? over : 0;
  }
}

// Pass 2 (after the exclusive scan of the overflow counts): offsets, records, overflow list.  rec is pre-zeroed.
__global__ __launch_bounds__(kThreads) void col16_encode_kernel(const int *__restrict__ ci, int nnz, int chunk0, int nchunks,
                                                                const int *__restrict__ base, const int *__restrict__ ovf_start,
                                                                int R, unsigned short *__restrict__ d16, int *__restrict__ rec,
                                                                int *__restrict__ ovf) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long c_ll = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave;
  if (c_ll >= nchunks) return;
  const int c = static_cast<int>(c_ll);
  const long long j0 = (static_cast<long long>(chunk0) + c) * kChunk + 4 * lane;
  const int bs = base[c];
  const int E = R - 4;
  int col[4];
  bool is_esc[4];
  int mine = 0;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    col[e] = (j0 + e < nnz) ? ci[j0 + e] : bs; // padding decodes to the base column (never read as a product)
    const long long d = static_cast<long long>(col[e]) - bs;
    is_esc[e] = d < 0 || d > 65534;
    mine += is_esc[e] ? 1 : 0;
  }
  // exclusive prefix of the lanes' escape counts
  int incl = mine;
#pragma unroll
  for (int o = 1; o < kWave; o <<= 1) {
    const int up = __shfl_up(incl, o, kWave);
    if (lane >= o) incl += up;
  }
  const int total = __shfl(incl, kWave - 1, kWave);
  int pos = incl - mine;
  int *r = rec + static_cast<size_t>(c) * R;
  const int o0 = ovf_start[c];
  unsigned short out[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    out[e] = 0xFFFF;
    if (is_esc[e]) {
      if (pos < E) r[4 + pos] = col[e];
      else ovf[o0 + pos - E] = col[e];
      ++pos;
    } else {
      out[e] = static_cast<unsigned short>(col[e] - bs);
    }
  }
  unsigned short *dst = d16 + static_cast<size_t>(c) * kChunk + 4 * lane; // (d16 holds whole chunks: the padding is written too)
#pragma unroll
  for (int e = 0; e < 4; ++e) dst[e] = out[e];
  if (lane == 0) {
    r[0] = bs;
    r[1] = total;
    r[2] = o0;
  }
}

// colindex samples for the stale-plan guard of the kernels that read the encoding instead of colindex (device_utils.hpp check_ci_guard)
__global__ __launch_bounds__(kWave) void col16_guard_kernel(const int *__restrict__ ci, int lo, int span, int *__restrict__ out) {
  out[threadIdx.x] = ci[lo + static_cast<int>(static_cast<long long>(threadIdx.x) * span / (kWave - 1))];
}

} // namespace

size_t col16_scan_bytes(int nchunks) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, static_cast<int *>(nullptr), static_cast<int *>(nullptr), 0,
                                static_cast<size_t>(nchunks) + 1, rocprim::plus<int>());
  return bytes;
}

void launch_col16_base(hipStream_t stream, const int *ci, int nnz, int chunk0, int nchunks, int *base, int *esc_count,
                       unsigned long long *stats) {
  if (nchunks <= 0) return;
  const int waves_per_block = kThreads / kWave;
  SPMV_ACC_LAUNCH(col16_base_kernel, dim3((nchunks + waves_per_block - 1) / waves_per_block), dim3(kThreads), 0, stream, ci, nnz,
                     chunk0, nchunks, base, esc_count, stats);
}

void launch_col16_overflow(hipStream_t stream, int *cnt, int nchunks, int E) {
  if (nchunks <= 0) return;
  SPMV_ACC_LAUNCH(col16_overflow_kernel, dim3((nchunks + kThreads - 1) / kThreads), dim3(kThreads), 0, stream, cnt, nchunks, E);
}

bool launch_col16_scan(hipStream_t stream, int nchunks, const int *esc_count, int *esc_start, void *tmp, size_t tmp_bytes) {
  // nchunks + 1 entries: esc_start[nchunks] = total (esc_count[nchunks] must be 0)
  return rocprim::exclusive_scan(tmp, tmp_bytes, esc_count, esc_start, 0, static_cast<size_t>(nchunks) + 1, rocprim::plus<int>(),
                                 stream) == hipSuccess;
}

void launch_col16_encode(hipStream_t stream, const int *ci, int nnz, int chunk0, int nchunks, const int *base, const int *ovf_start,
                         int R, unsigned short *d16, int *rec, int *ovf) {
  if (nchunks <= 0) return;
  const int waves_per_block = kThreads / kWave;
  SPMV_ACC_LAUNCH(col16_encode_kernel, dim3((nchunks + waves_per_block - 1) / waves_per_block), dim3(kThreads), 0, stream, ci, nnz,
                     chunk0, nchunks, base, ovf_start, R, d16, rec, ovf);
}

void launch_col16_guard(hipStream_t stream, const int *ci, int lo, int span, int *out) {
  SPMV_ACC_LAUNCH(col16_guard_kernel, dim3(1), dim3(kWave), 0, stream, ci, lo, span, out);
}

} // namespace spmv_acc
