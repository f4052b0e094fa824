// k_analyze.hip -- DEVICE form of the csr-adaptive-plus row-block analysis.
//
// Reference role: csr_adaptive_plus_analyze_imp (hip-csr-adaptive-plus/csr_adaptive_plus_analyze.cpp:13-98), a
// single sequential HOST loop over all m rows followed by two blocking uploads, repeated on every SpMV call
// (csr_adaptive_plus_spmv.cpp:18-41).  The loop is a greedy segmentation -- "open a block at row s, close it at the
// first row where it holds >= MIN_NNZ non-zeros or THREADS/VEC rows" -- so the block that opens at s ends at a row
// next(s) that depends on s alone.  That turns the scan into list traversal, which parallelises:
//   1. next[s] for every row s           (one bounded binary search on rowptr per row)
//   2. which rows open a block           (the orbit of row 0 under next, by pointer jumping: log2(blocks) rounds)
//   3. entries each block emits          (1, or for a long row: its slice count), exclusive scan -> positions
//   4. write break_points / first_block_of_row
// The tables are bit-identical to the host form (tests pin both against the compiled reference).
#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

// closing row of the block that opens at row s (host form: config.cpp::plus_analyze_host)
__global__ __launch_bounds__(256) void next_kernel(const int *__restrict__ rp, int m, int min_nnz, int cap,
                                                   int *__restrict__ next) {
  const long long s_ll = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (s_ll > m) return;
  const int s = static_cast<int>(s_ll);
  if (s == m) {
    next[m] = m; // fixed point: the chain ends here
    return;
  }
  const long long target = static_cast<long long>(rp[s]) + min_nnz;
  const long long wend_ll = static_cast<long long>(s) + cap;
  const int wend = wend_ll < m ? static_cast<int>(wend_ll) : m; // last rowptr index a block opened at s may reach
  // first q in [s+1, wend] with rp[q] >= target  (rp is non-decreasing)
  int lo = s + 1, hi = wend + 1;
  while (lo < hi) {
    const int mid = lo + (hi - lo) / 2;
    if (rp[mid] < target) lo = mid + 1; else hi = mid;
  }
  // lo <= wend: the block closes on its non-zero count after row lo-1; otherwise on the row cap / matrix end
  next[s] = (lo <= wend) ? lo : wend;
}

// one pointer-jumping round: rows reached so far mark where they jump to; every row squares its jump
__global__ __launch_bounds__(256) void jump_kernel(int m, const int *__restrict__ jin, int *__restrict__ jout,
                                                   unsigned char *__restrict__ reach) {
  const long long s_ll = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (s_ll > m) return;
  const int s = static_cast<int>(s_ll);
  const int j = jin[s];
  if (reach[s]) reach[j] = 1; // idempotent store: concurrent writers agree
  jout[s] = jin[j];
}

struct BlockShape {
  int close_row;  // last row of the block
  bool nnz_close; // closed because it reached MIN_NNZ (else: row cap / end of matrix)
  bool is_long;   // closing row gets dedicated slices
  bool alone;     // nothing but the closing row's non-zeros in the block
  int slices;
};

__device__ __forceinline__ BlockShape block_shape(const int *__restrict__ rp, int s, int nxt, int min_nnz) {
  BlockShape b;
  b.close_row = nxt - 1;
  const long long len = static_cast<long long>(rp[nxt]) - rp[b.close_row];
  const long long total = static_cast<long long>(rp[nxt]) - rp[s];
  b.nnz_close = total >= min_nnz;
  b.is_long = b.nnz_close && len >= 2LL * min_nnz;
  b.alone = total == len;
  b.slices = b.is_long ? static_cast<int>(len / (2LL * min_nnz)) : 0;
  return b;
}

__global__ __launch_bounds__(256) void count_kernel(const int *__restrict__ rp, int m, int min_nnz,
                                                    const int *__restrict__ next,
                                                    const unsigned char *__restrict__ reach, int *__restrict__ cnt) {
  const long long s_ll = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (s_ll > m) return;
  const int s = static_cast<int>(s_ll);
  int c = 0;
  if (s < m && reach[s]) {
    const BlockShape b = block_shape(rp, s, next[s], min_nnz);
    c = b.is_long ? b.slices - (b.alone ? 1 : 0) + 1 : 1;
  }
  cnt[s] = c;
}

// pass 0: break points + the first_block_of_row marks of blocks that closed on the row cap
// pass 1: the marks of long rows (they overwrite a cap mark on the same row, as in the sequential order)
template <int PASS>
__global__ __launch_bounds__(256) void emit_kernel(const int *__restrict__ rp, int m, int min_nnz,
                                                   const int *__restrict__ next,
                                                   const unsigned char *__restrict__ reach,
                                                   const int *__restrict__ base, int *__restrict__ bp,
                                                   int *__restrict__ fbr) {
  const long long s_ll = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (s_ll >= m) return;
  const int s = static_cast<int>(s_ll);
  if (!reach[s]) return;
  const BlockShape b = block_shape(rp, s, next[s], min_nnz);
  int idx = base[s]; // position of this block's own start entry (value s)
  if (PASS == 0) {
    if (s == 0) bp[0] = 0;
    if (b.is_long) {
      for (int k = 0; k < b.slices; ++k) {
        if (!(k == 0 && b.alone)) bp[++idx] = b.close_row;
      }
      bp[++idx] = b.close_row + 1;
    } else {
      bp[++idx] = b.close_row + 1;
      if (!b.nnz_close) fbr[b.close_row + 1] = 2 * idx;
    }
  } else if (b.is_long) {
    fbr[b.close_row] = 2 * (b.alone ? idx : idx + 1) + 1;
  }
}

inline int grid_for(long long n) { return static_cast<int>((n + 255) / 256); }

} // namespace

size_t plus_analyze_device_workspace_bytes(int m) {
  const size_t n = static_cast<size_t>(m) + 1;
  size_t scan_bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, scan_bytes, static_cast<int *>(nullptr), static_cast<int *>(nullptr), 0, n,
                                rocprim::plus<int>());
  // next, two jump buffers, counts, positions (4 B each) + reach (1 B) + scan scratch, each 256-B aligned
  return 5 * ((4 * n + 255) / 256 * 256) + ((n + 255) / 256 * 256) + ((scan_bytes + 255) / 256 * 256) + 256;
}

// Enqueues steps 1-3 on `stream`.  After the stream has run, d_total[0] = number of row blocks and the
// workspace holds what plus_analyze_device_emit needs.  Returns false on a rocPRIM error.
bool plus_analyze_device_count(hipStream_t stream, const int *rp, int m, int min_nnz, int threads_per_block,
                               int vec_size, void *workspace, int *d_total) {
  const size_t n = static_cast<size_t>(m) + 1;
  const size_t a4 = (4 * n + 255) / 256 * 256, a1 = (n + 255) / 256 * 256;
  char *w = static_cast<char *>(workspace);
  int *next = reinterpret_cast<int *>(w);
  int *ja = reinterpret_cast<int *>(w + a4);
  int *jb = reinterpret_cast<int *>(w + 2 * a4);
  int *cnt = reinterpret_cast<int *>(w + 3 * a4);
  int *base = reinterpret_cast<int *>(w + 4 * a4);
  unsigned char *reach = reinterpret_cast<unsigned char *>(w + 5 * a4);
  void *scan_tmp = w + 5 * a4 + a1;
  const int cap = threads_per_block / vec_size;
  const int g = grid_for(static_cast<long long>(n));

  SPMV_ACC_LAUNCH(next_kernel, dim3(g), dim3(256), 0, stream, rp, m, min_nnz, cap, next);
  (void)hipMemsetAsync(reach, 0, n, stream);
  (void)hipMemsetAsync(reach, 1, 1, stream); // row 0 opens the first block
  (void)hipMemcpyAsync(ja, next, 4 * n, hipMemcpyDeviceToDevice, stream);
  // after k rounds every start within 2^k blocks of row 0 is marked; a block holds >= 1 row, so m blocks at most
  int rounds = 1;
  while ((1LL << rounds) < static_cast<long long>(m) + 1) ++rounds;
  for (int k = 0; k < rounds; ++k) {
    SPMV_ACC_LAUNCH(jump_kernel, dim3(g), dim3(256), 0, stream, m, ja, jb, reach);
    int *t = ja;
    ja = jb;
    jb = t;
  }
  SPMV_ACC_LAUNCH(count_kernel, dim3(g), dim3(256), 0, stream, rp, m, min_nnz, next, reach, cnt);
  size_t scan_bytes = 0;
  (void)rocprim::exclusive_scan(nullptr, scan_bytes, cnt, base, 0, n, rocprim::plus<int>(), stream);
  if (rocprim::exclusive_scan(scan_tmp, scan_bytes, cnt, base, 0, n, rocprim::plus<int>(), stream) != hipSuccess)
    return false;
  // cnt[m] = 0, so base[m] is the total number of emitted entries = number of blocks
  (void)hipMemcpyAsync(d_total, base + m, sizeof(int), hipMemcpyDeviceToDevice, stream);
  return true;
}

// Step 4: d_bp needs total + 1 entries, d_fbr m + 1 entries (zeroed here).
void plus_analyze_device_emit(hipStream_t stream, const int *rp, int m, int min_nnz, const void *workspace, int *d_bp,
                              int *d_fbr) {
  const size_t n = static_cast<size_t>(m) + 1;
  const size_t a4 = (4 * n + 255) / 256 * 256;
  const char *w = static_cast<const char *>(workspace);
  const int *next = reinterpret_cast<const int *>(w);
  const int *base = reinterpret_cast<const int *>(w + 4 * a4);
  const unsigned char *reach = reinterpret_cast<const unsigned char *>(w + 5 * a4);
  (void)hipMemsetAsync(d_fbr, 0, 4 * n, stream);
  if (m == 0) {
    (void)hipMemsetAsync(d_bp, 0, sizeof(int), stream);
    return;
  }
  const int g = grid_for(m);
  SPMV_ACC_LAUNCH((emit_kernel<0>), dim3(g), dim3(256), 0, stream, rp, m, min_nnz, next, reach, base, d_bp, d_fbr);
  SPMV_ACC_LAUNCH((emit_kernel<1>), dim3(g), dim3(256), 0, stream, rp, m, min_nnz, next, reach, base, d_bp, d_fbr);
}

} // namespace spmv_acc
