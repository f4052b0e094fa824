// k_guard.hip -- OPT-IN full row-pointer check (tunable guard_full): every SpMV call re-reads ALL of rowptr and compares a 64-bit
// digest with the one taken when the plan was built.
//
// Reference role: none needed there -- the reference recomputes its preprocessing from the caller's arrays on every call
// (flat.cpp:39-44, line_enhance_spmv.cpp), so it can never act on tables of a matrix that has since changed.  Plans here are kept,
// and the stale-plan guard every kernel carries (device_utils.hpp::check_plan_guard) compares 64 strided samples of rowptr: an
// in-place edit of the structure that leaves all 64 untouched goes unnoticed ("What's weak" 10).  This mode closes
// that window for callers who edit structures in place and cannot announce it: 4 * (m + 1) bytes more per SpMV (3-6 % of the
// traffic of the sweep stand-ins) and two small launches, instead of a rebuild of every table per call.
//
// digest = sum over i of rowptr[i] * odd(i)   (mod 2^64),   odd(i) = (2 i + 1) * 0x9E3779B97F4A7C15
// Every weight is odd, hence invertible mod 2^64: a change of ONE entry always changes the digest; several changes cancel only by
// accident (2^-64).  A sum, so the order in which the parts are added does not matter: every workgroup writes ONE word, a second
// one-workgroup kernel adds the words up and compares.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

__global__ __launch_bounds__(kThreads) void rowptr_digest_kernel(const int *__restrict__ rp, long long count,
                                                                 unsigned long long *__restrict__ part) {
  unsigned long long h = 0;
  const long long stride = static_cast<long long>(gridDim.x) * kThreads;
  for (long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x; i < count; i += stride)
    h += static_cast<unsigned long long>(static_cast<unsigned>(rp[i])) * ((2ULL * static_cast<unsigned long long>(i) + 1ULL) * 0x9E3779B97F4A7C15ULL);
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) h += __shfl_xor(h, off, kWave);
  __shared__ unsigned long long wave_part[kThreads / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) wave_part[threadIdx.x / kWave] = h;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0;
#pragma unroll
    for (int w = 0; w < kThreads / kWave; ++w) s += wave_part[w];
    part[blockIdx.x] = s; // one word per workgroup, no atomics: 8192 same-address atomics cost 100 us, this costs nothing
  }
}

// one workgroup: add the parts up; then either hand the digest out (plan build) or compare it with the plan's and raise the plan's
// sticky flag (pinned host memory, like check_plan_guard)
__global__ __launch_bounds__(kThreads) void rowptr_verdict_kernel(const unsigned long long *__restrict__ part, int parts,
                                                                  unsigned long long expected, int *__restrict__ stale,
                                                                  unsigned long long *__restrict__ digest_out) {
  unsigned long long h = 0;
  for (int i = threadIdx.x; i < parts; i += kThreads) h += part[i];
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) h += __shfl_xor(h, off, kWave);
  __shared__ unsigned long long wave_part[kThreads / kWave];
  if ((threadIdx.x & (kWave - 1)) == 0) wave_part[threadIdx.x / kWave] = h;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0;
#pragma unroll
    for (int w = 0; w < kThreads / kWave; ++w) s += wave_part[w];
    if (digest_out != nullptr) *digest_out = s;
    else if (s != expected && stale != nullptr) __hip_atomic_store(stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

} // namespace

int rowptr_digest_parts(int m) {
  const long long count = static_cast<long long>(m) + 1;
  long long blocks = (count + kThreads * 16 - 1) / (kThreads * 16); // 16 entries per lane, at most kDigestMaxParts workgroups
  if (blocks > kDigestMaxParts) blocks = kDigestMaxParts;
  return blocks < 1 ? 1 : static_cast<int>(blocks);
}

void launch_rowptr_digest(hipStream_t stream, const int *rp, int m, unsigned long long *part) {
  SPMV_ACC_LAUNCH(rowptr_digest_kernel, dim3(rowptr_digest_parts(m)), dim3(kThreads), 0, stream, rp, static_cast<long long>(m) + 1, part);
}

void launch_rowptr_verdict(hipStream_t stream, const unsigned long long *part, int m, unsigned long long expected, int *stale,
                           unsigned long long *digest_out) {
  SPMV_ACC_LAUNCH(rowptr_verdict_kernel, dim3(1), dim3(kThreads), 0, stream, part, rowptr_digest_parts(m), expected, stale, digest_out);
}

} // namespace spmv_acc
