// last_arriver_bench.hip -- round 4: what would a SINGLE-LAUNCH fold of flat's carries cost?  Today the rows a tile cuts are folded by a second launch
// (3-4 us).  The alternative: every tile publishes its part, bumps an arrival counter of the row, and the LAST tile to arrive adds the parts in tile order
// (deterministic: the order of the sum does not depend on who is last).  On MI355X the parts cross eight L2s, so the publish / consume pair needs
// agent-scope semantics.  This measures a streaming kernel (each block reads 16 KB, like a tile) in three forms:
//   mode 0: plain -- every block writes its partial, nothing else                                   (+ mode 0b: a second tiny launch folds groups of G)
//   mode 1: __threadfence() + relaxed atomicAdd on the group's counter; the last arriver __threadfence()s and reads the parts with plain loads
//   mode 2: parts stored and loaded with agent-scope atomics (sc1 write-through / L2-bypassing loads), release / acquire on the counter only through
//           s_waitcnt (no L2 write-back instruction)
// and checks every group sum of every repetition (a stale part would show as a wrong sum).
// Usage: last_arriver_bench [MB=600] [G=3]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double2v __attribute__((ext_vector_type(2)));
constexpr int kBlockBytes = 16384;

__device__ __forceinline__ double block_sum(const double *__restrict__ a, long long b) {
  const double2v *p = reinterpret_cast<const double2v *>(a) + b * (kBlockBytes / 16);
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double2v u = __builtin_nontemporal_load(p + k * 256 + threadIdx.x);
    s += u.x + u.y;
  }
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  return part[0] + part[1] + part[2] + part[3];
}

template <int MODE>
__global__ __launch_bounds__(256) void tile(const double *__restrict__ a, double stamp, double *parts, int *cnt, double *out, int G) {
  const long long b = blockIdx.x;
  const double s = block_sum(a, b) + stamp + static_cast<double>(b % 7);
  if (threadIdx.x != 0) return;
  const long long g = b / G;
  if (MODE == 0) {
    parts[b] = s;
    return;
  }
  int old;
  if (MODE == 1) {
    parts[b] = s;
    __threadfence();
    old = atomicAdd(cnt + g, 1);
  } else {
    __hip_atomic_store(parts + b, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0); // the write-through store has been acknowledged before the counter moves
    old = __hip_atomic_fetch_add(cnt + g, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (old == G - 1) { // the last arriver folds the group in index order
    if (MODE == 1) __threadfence();
    double sum = 0.0;
    for (int k = 0; k < G; ++k)
      sum += MODE == 1 ? parts[g * G + k] : __hip_atomic_load(parts + g * G + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    out[g] = sum;
    if (MODE == 1) cnt[g] = 0;
    else __hip_atomic_store(cnt + g, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // ready for the next launch (stream order publishes it)
  }
}
__global__ void fold(const double *parts, double *out, long long groups, int G) {
  const long long g = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x;
  if (g >= groups) return;
  double sum = 0.0;
  for (int k = 0; k < G; ++k) sum += parts[g * G + k];
  out[g] = sum;
}

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                  \
      return 1;                                                                   \
    }                                                                             \
  } while (0)

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const double mb = argc > 1 ? atof(argv[1]) : 600.0;
  const int G = argc > 2 ? atoi(argv[2]) : 3;
  const long long nblocks = static_cast<long long>(mb * 1e6 / kBlockBytes) / G * G, groups = nblocks / G;
  double *a = nullptr, *parts = nullptr, *out = nullptr;
  int *cnt = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void **>(&a), nblocks * kBlockBytes));
  CHECK(hipMalloc(reinterpret_cast<void **>(&parts), nblocks * sizeof(double)));
  CHECK(hipMalloc(reinterpret_cast<void **>(&out), groups * sizeof(double)));
  CHECK(hipMalloc(reinterpret_cast<void **>(&cnt), groups * sizeof(int)));
  CHECK(hipMemset(a, 0, nblocks * kBlockBytes));
  CHECK(hipMemset(cnt, 0, groups * sizeof(int)));
  hipEvent_t t0, t1;
  CHECK(hipEventCreate(&t0));
  CHECK(hipEventCreate(&t1));
  std::vector<double> h(groups);
  std::printf("%lld blocks of 16 KB (%.0f MB), groups of %d\n", nblocks, nblocks * 16384e-6, G);
  const char *names[] = {"plain (parts only)", "plain + fold launch", "__threadfence + atomicAdd, last arriver folds", "agent-scope stores / loads, s_waitcnt, last arriver folds"};
  double stamp = 0.0;
  for (int mode = 0; mode < 4; ++mode) {
    float sum_ms = 0.f, best = 1e30f;
    long long wrong = 0;
    const int reps = 40;
    for (int r = -5; r < reps; ++r) {
      stamp += 1.0;
      CHECK(hipMemsetAsync(out, 0, groups * sizeof(double), nullptr));
      CHECK(hipDeviceSynchronize());
      CHECK(hipEventRecord(t0, nullptr));
      if (mode <= 1) hipLaunchKernelGGL(tile<0>, dim3(static_cast<unsigned>(nblocks)), dim3(256), 0, nullptr, a, stamp, parts, cnt, out, G);
      if (mode == 1) hipLaunchKernelGGL(fold, dim3(static_cast<unsigned>((groups + 255) / 256)), dim3(256), 0, nullptr, parts, out, groups, G);
      if (mode == 2) hipLaunchKernelGGL(tile<1>, dim3(static_cast<unsigned>(nblocks)), dim3(256), 0, nullptr, a, stamp, parts, cnt, out, G);
      if (mode == 3) hipLaunchKernelGGL(tile<2>, dim3(static_cast<unsigned>(nblocks)), dim3(256), 0, nullptr, a, stamp, parts, cnt, out, G);
      CHECK(hipEventRecord(t1, nullptr));
      CHECK(hipDeviceSynchronize());
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, t0, t1));
      if (r >= 0) {
        sum_ms += ms;
        best = ms < best ? ms : best;
      }
      if (mode >= 1) {
        CHECK(hipMemcpy(h.data(), out, groups * sizeof(double), hipMemcpyDeviceToHost));
        for (long long g = 0; g < groups; ++g) {
          double want = 0.0;
          for (int k = 0; k < G; ++k) want += stamp + static_cast<double>((g * G + k) % 7);
          wrong += h[g] != want;
        }
      }
    }
    std::printf("mode %d  %-58s: mean %7.1f us  min %7.1f us   wrong group sums over %d launches: %lld\n", mode, names[mode], sum_ms / reps * 1e3,
                best * 1e3, reps + 5, wrong);
  }
  return 0;
}
