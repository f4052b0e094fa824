// gather_request_size_bench.hip -- round 5: how many BYTES does one L2-missing 8-byte gather move over the fabric, and does any load flavour or memory
// type make it fewer?  (profiles/r05_fem_kernels_pmc.md: in the SpMV kernels every L2-missing read is a 128-B request -- a far gather costs a whole line
// for 8 useful bytes, and the card's "54 G random gathers/s" of round 1 is 54 G x 128 B = 6.9 TB/s, i.e. the fabric's byte rate, not a request rate.)
// Random 8-B gathers from a table far beyond the L2s with
//   load flavour:  0 plain   1 nt   2 sc1 (agent-scope relaxed atomic load)   3 sc0 sc1 (system-scope relaxed atomic load)   4 asm sc0 sc1 nt
//   table memory:  0 hipMalloc   1 hipExtMallocWithFlags(hipDeviceMallocUncached)   2 hipExtMallocWithFlags(hipDeviceMallocFinegrained)
// Run plainly for the rates; under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum` for the
// request sizes (kernel k<MODE, MEM>: one dispatch per pair when argv[3] = 1).
// Usage: gather_request_size_bench <table_MB> <Mgathers> [one_dispatch_each]      (hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int int4v __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ double gather(const double *x, int i) {
  if (MODE == 1) return __builtin_nontemporal_load(x + i);
  if (MODE == 2) return __hip_atomic_load(x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 3) return __hip_atomic_load(x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (MODE == 4) {
    double v;
    const double *p = x + i;
    asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1 nt\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
  }
  return x[i];
}

template <int MODE, int MEM> __global__ __launch_bounds__(256) void k(const int *idx, long long n4, const double *x, double *out) {
  const long long t = static_cast<long long>(blockIdx.x) * 512 + threadIdx.x;
  double s = 0;
  int4v a, b;
  const bool fa = t < n4, fb = t + 256 < n4;
  if (fa) a = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(idx) + t);
  if (fb) b = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(idx) + t + 256);
  if (fa) s += gather<MODE>(x, a.x) + gather<MODE>(x, a.y) + gather<MODE>(x, a.z) + gather<MODE>(x, a.w);
  if (fb) s += gather<MODE>(x, b.x) + gather<MODE>(x, b.y) + gather<MODE>(x, b.z) + gather<MODE>(x, b.w);
  if (s == 123.456) out[0] = s; // keep the loads alive
}

template <int MODE, int MEM> float run(int reps, int grid, const int *idx, long long n4, const double *x, double *out, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, MEM>), dim3(grid), dim3(256), 0, 0, idx, n4, x, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  const long long table_mb = argc > 1 ? atoll(argv[1]) : 512;
  const long long n = (argc > 2 ? atoll(argv[2]) : 64) * 1000000LL / 4 * 4;
  const int reps = (argc > 3 && atoi(argv[3]) == 1) ? 1 : 5;
  const long long elems = table_mb * 1000000LL / 8;
  std::vector<int> h(n);
  unsigned long long st = 88172645463325252ULL;
  for (long long i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = static_cast<int>(st % elems); }
  int *d_idx; double *d_out; double *tab[3] = {nullptr, nullptr, nullptr};
  hipMalloc(&d_idx, n * 4); hipMalloc(&d_out, 8);
  hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMalloc(&tab[0], elems * 8);
  if (hipExtMallocWithFlags(reinterpret_cast<void **>(&tab[1]), elems * 8, hipDeviceMallocUncached) != hipSuccess) { tab[1] = nullptr; (void)hipGetLastError(); }
  if (hipExtMallocWithFlags(reinterpret_cast<void **>(&tab[2]), elems * 8, hipDeviceMallocFinegrained) != hipSuccess) { tab[2] = nullptr; (void)hipGetLastError(); }
  for (auto t : tab) if (t) hipMemset(t, 0, elems * 8);
  hipDeviceSynchronize();
  const long long n4 = n / 4;
  const int grid = static_cast<int>((n4 + 511) / 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *flav[5] = {"plain", "nt", "sc1 (agent)", "sc0 sc1 (system)", "asm sc0 sc1 nt"};
  const char *mem[3] = {"hipMalloc", "uncached", "fine-grained"};
#define RUN(MODE, MEM)                                                                                                                     \
  if (tab[MEM]) {                                                                                                                          \
    if (reps > 1) (void)run<MODE, MEM>(1, grid, d_idx, n4, tab[MEM], d_out, e0, e1);                                                       \
    const float ms = run<MODE, MEM>(reps, grid, d_idx, n4, tab[MEM], d_out, e0, e1);                                                       \
    printf("table %lld MB (%s), %lld M gathers, %-18s k<%d, %d>: %8.1f us  %6.1f G gathers/s\n", table_mb, mem[MEM], n / 1000000, flav[MODE], \
           MODE, MEM, ms * 1e3, n / (ms * 1e-3) / 1e9);                                                                                    \
  }
  RUN(0, 0) RUN(1, 0) RUN(2, 0) RUN(3, 0) RUN(4, 0)
  RUN(0, 1) RUN(1, 1) RUN(3, 1)
  RUN(0, 2) RUN(1, 2) RUN(3, 2)
  return 0;
}
