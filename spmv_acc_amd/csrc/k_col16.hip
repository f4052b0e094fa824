// k_col16.hip -- OPT-IN 16-bit column encoding for the flat family (tunable `col16`, off by default).
//
// No reference counterpart: the reference streams 4-byte colindex entries (hip-flat/flat_imp_one_pass.hpp:16-77 reads
// 12 B per non-zero: 8 value + 4 column).  On MI355X the tile kernels run at the device's streaming-copy rate on matrices
// whose gathers hit in L2, so the only lever left is moving fewer bytes.  Where columns are local -- FEM / banded matrices:
// the non-zeros of a few neighbouring rows lie within a few hundred columns of each other -- a column can be stored as a
// 16-bit offset from a per-chunk base:
//     chunk  = 256 consecutive non-zeros (one wavefront's 16-B-equivalent step: 64 lanes x 4)
//     base[c]   = median of the chunk's 64 lane minima - 32767, clamped at 0      (robust against far columns)
//     d16[j]    = colindex[j] - base[j / 256]          if that fits in [0, 65534]
//               = 0xFFFF (escape)                       otherwise; the column then sits in esc_cols, in non-zero order,
//     esc_start[c] = index of the chunk's first escape in esc_cols
// 4 B/nnz of column stream become 2 B/nnz + 8 B/chunk + 4 B per escape: the 12 B/nnz stream drops to ~10.1 B/nnz at 2 % far
// columns.  The price is that the PLAN now holds a derived copy of colindex: a caller that edits colindex in place (same
// rowptr) must call spmv_acc_release_plans -- which is why this is opt-in and never the default (plans otherwise survive
// in-place edits of values and of column indices that keep rowptr).
#include <climits>

#include <rocprim/device/device_scan.hpp>

#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

constexpr int kChunk = kCol16Chunk;

// One wavefront per chunk.  Pass 1: base + escape count.
__global__ __launch_bounds__(kThreads) void col16_base_kernel(const int *__restrict__ ci, int nnz, int nchunks,
                                                              int *__restrict__ base, int *__restrict__ esc_count) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long c_ll = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave;
  if (c_ll >= nchunks) return; // wave-uniform
  const int c = static_cast<int>(c_ll);
  const int j0 = c * kChunk + 4 * lane;
  int col[4];
  int mine = INT_MAX;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    col[e] = (j0 + e < nnz) ? ci[j0 + e] : INT_MAX;
    mine = col[e] < mine ? col[e] : mine;
  }
  // rank of this lane's minimum among the 64 lane minima (ties broken by lane id): the lane of rank 32 holds the median
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
  }
}

// Pass 2 (after the exclusive scan of esc_count into esc_start): offsets + escape list.
__global__ __launch_bounds__(kThreads) void col16_encode_kernel(const int *__restrict__ ci, int nnz, int nchunks,
                                                                const int *__restrict__ base,
                                                                const int *__restrict__ esc_start,
                                                                unsigned short *__restrict__ d16,
                                                                int *__restrict__ esc_cols) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long c_ll = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave;
  if (c_ll >= nchunks) return;
  const int c = static_cast<int>(c_ll);
  const int j0 = c * kChunk + 4 * lane;
  const int bs = base[c];
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
  int pos = esc_start[c] + incl - mine;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    unsigned short out = 0xFFFF;
    if (is_esc[e]) esc_cols[pos++] = col[e];
    else out = static_cast<unsigned short>(col[e] - bs);
    d16[j0 + e] = out; // d16 is allocated with nchunks * 256 entries: the padding is written too
  }
}

} // namespace

size_t col16_scan_bytes(int nchunks) {
  size_t bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, bytes, static_cast<int *>(nullptr), static_cast<int *>(nullptr), 0,
                                static_cast<size_t>(nchunks) + 1, rocprim::plus<int>());
  return bytes;
}

void launch_col16_base(hipStream_t stream, const int *ci, int nnz, int nchunks, int *base, int *esc_count) {
  if (nchunks <= 0) return;
  const int waves_per_block = kThreads / kWave;
  SPMV_ACC_LAUNCH(col16_base_kernel, dim3((nchunks + waves_per_block - 1) / waves_per_block), dim3(kThreads), 0, stream, ci, nnz,
                     nchunks, base, esc_count);
}

bool launch_col16_scan(hipStream_t stream, int nchunks, const int *esc_count, int *esc_start, void *tmp, size_t tmp_bytes) {
  // nchunks + 1 entries: esc_start[nchunks] = total number of escapes (esc_count[nchunks] must be 0)
  return rocprim::exclusive_scan(tmp, tmp_bytes, esc_count, esc_start, 0, static_cast<size_t>(nchunks) + 1, rocprim::plus<int>(),
                                 stream) == hipSuccess;
}

void launch_col16_encode(hipStream_t stream, const int *ci, int nnz, int nchunks, const int *base, const int *esc_start,
                         unsigned short *d16, int *esc_cols) {
  if (nchunks <= 0) return;
  const int waves_per_block = kThreads / kWave;
  SPMV_ACC_LAUNCH(col16_encode_kernel, dim3((nchunks + waves_per_block - 1) / waves_per_block), dim3(kThreads), 0, stream, ci, nnz,
                     nchunks, base, esc_start, d16, esc_cols);
}

} // namespace spmv_acc
