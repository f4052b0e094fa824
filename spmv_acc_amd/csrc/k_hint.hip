// k_hint.hip -- gather hints: which x[] gathers of a matrix are worth keeping in L2?
//
// No reference counterpart (the reference gathers x with plain loads everywhere, e.g. hip-flat/flat_imp_one_pass.hpp:35-39).
// On MI355X a gather that misses L2 costs one fabric request and brings a 128-B line into a 4 MB L2; the card serves ~54-58 G
// such requests per second whatever the load flavour (profiles/r01_gather_microbench.txt).  Where the columns of a matrix follow a
// power law (graphs: R-MAT, web / social matrices) a small set of x lines takes a large share of the gathers, but the lines the
// other gathers bring in push it out of L2 again.  Issuing just those other gathers non-temporal keeps the hot set resident:
// profiles/probes/micro/skewed_gather_bench.hip measures 66 -> 74-76 G gathers/s on R-MAT scale-25 columns (all gathers non-temporal: 43).
//
// The plan therefore takes a census of the matrix' columns once:
//   1. hint_census_kernel: a uniform sample of up to kHintSamples non-zeros; one atomic per sample on the counter of its x line
//      (16 columns = 128 B, the L2 line);
//   2. hint_hist_kernel: lines and sampled hits per count value (4096 bins, the last one open-ended); the host picks the count
//      threshold T whose lines (count >= T) fit the budget (tunable hint_budget_kb) and learns what share of the gathers they take;
//   3. hint_bits_kernel: one bit per non-zero, set where its line's count is below T ("cold").
// The bits only steer a cache policy.  They are derived from colindex, but a stale bit cannot change a result, so -- unlike the
// 16-bit column encoding -- the plan needs no guard for them and they are on by default wherever the engine's timing says they pay.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

__device__ __forceinline__ unsigned mix32(unsigned a) {
  a ^= a >> 16;
  a *= 0x7feb352dU;
  a ^= a >> 15;
  a *= 0x846ca68bU;
  a ^= a >> 16;
  return a;
}

__global__ __launch_bounds__(256) void hint_census_kernel(const int *__restrict__ ci, int nnz, int stride, int samples, int ncols,
                                                          unsigned *__restrict__ counts) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < samples; i += gridDim.x * 256LL) {
    long long j = i * stride + static_cast<long long>(mix32(static_cast<unsigned>(i)) % static_cast<unsigned>(stride));
    if (j >= nnz) j = nnz - 1;
    const int c = ci[j];
    if (c >= 0 && c < ncols) atomicAdd(&counts[c >> kHintLineShift], 1u);
  }
}

__global__ __launch_bounds__(256) void hint_hist_kernel(const unsigned *__restrict__ counts, int nlines, unsigned *__restrict__ hist_lines,
                                                        unsigned long long *__restrict__ hist_hits) {
  __shared__ unsigned l_lines[kHintBins];
  __shared__ unsigned l_hits[kHintBins]; // (a block sees at most 256 * 64 lines: sums stay far below 2^32 unless counts do not)
  for (int b = threadIdx.x; b < kHintBins; b += 256) {
    l_lines[b] = 0;
    l_hits[b] = 0;
  }
  __syncthreads();
  unsigned long long big_hits = 0; // hits of the open-ended last bin, accumulated without the 32-bit LDS cell
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < nlines; i += gridDim.x * 256LL) {
    const unsigned c = counts[i];
    if (c == 0) continue;
    if (c >= kHintBins - 1) {
      atomicAdd(&l_lines[kHintBins - 1], 1u);
      big_hits += c;
    } else {
      atomicAdd(&l_lines[c], 1u);
      atomicAdd(&l_hits[c], c);
    }
  }
  if (big_hits) atomicAdd(&hist_hits[kHintBins - 1], big_hits);
  __syncthreads();
  for (int b = threadIdx.x; b < kHintBins; b += 256) {
    if (l_lines[b]) atomicAdd(&hist_lines[b], l_lines[b]);
    if (l_hits[b]) atomicAdd(&hist_hits[b], static_cast<unsigned long long>(l_hits[b]));
  }
}

// one byte (8 non-zeros) per lane
__global__ __launch_bounds__(256) void hint_bits_kernel(const int *__restrict__ ci, int nnz, int ncols, const unsigned *__restrict__ counts,
                                                        unsigned threshold, unsigned char *__restrict__ bits, long long nbytes) {
  for (long long b = blockIdx.x * 256LL + threadIdx.x; b < nbytes; b += gridDim.x * 256LL) {
    const long long j0 = b * 8;
    unsigned byte = 0;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      if (j0 + e < nnz) {
        const int c = ci[j0 + e];
#ifdef SPMV_ACC_HINT_BY_POSITION // A/B builds only: cold = a column far (> 4096) from both its neighbours in the stream, whatever its popularity
        const long long j = j0 + e;
        const int before = j > 0 ? ci[j - 1] : c, after = j + 1 < nnz ? ci[j + 1] : c;
        const bool cold = (c < 0 || c >= ncols) || ((c > before ? c - before : before - c) > 4096 && (c > after ? c - after : after - c) > 4096);
        (void)counts; (void)threshold;
#else
        const bool cold = (c < 0 || c >= ncols) ? true : counts[c >> kHintLineShift] < threshold;
#endif
        byte |= cold ? (1u << e) : 0u;
      }
    }
    bits[b] = static_cast<unsigned char>(byte);
  }
}

} // namespace

void launch_hint_census(hipStream_t stream, const int *ci, int nnz, int ncols, int stride, int samples, unsigned *counts) {
  if (samples <= 0) return;
  const long long blocks = (static_cast<long long>(samples) + 255) / 256;
  SPMV_ACC_LAUNCH(hint_census_kernel, dim3(static_cast<unsigned>(blocks < 65536 ? blocks : 65536)), dim3(256), 0, stream, ci, nnz, stride,
                     samples, ncols, counts);
}

void launch_hint_hist(hipStream_t stream, const unsigned *counts, int nlines, unsigned *hist_lines, unsigned long long *hist_hits) {
  if (nlines <= 0) return;
  const long long blocks = (static_cast<long long>(nlines) + 256 * 64 - 1) / (256 * 64);
  SPMV_ACC_LAUNCH(hint_hist_kernel, dim3(static_cast<unsigned>(blocks)), dim3(256), 0, stream, counts, nlines, hist_lines, hist_hits);
}

void launch_hint_bits(hipStream_t stream, const int *ci, int nnz, int ncols, const unsigned *counts, unsigned threshold, unsigned char *bits) {
  const long long nbytes = (static_cast<long long>(nnz) + 7) / 8;
  if (nbytes <= 0) return;
  const long long blocks = (nbytes + 255) / 256;
  SPMV_ACC_LAUNCH(hint_bits_kernel, dim3(static_cast<unsigned>(blocks < 262144 ? blocks : 262144)), dim3(256), 0, stream, ci, nnz, ncols,
                     counts, threshold, bits, nbytes);
}

} // namespace spmv_acc
