// gather_bench.hip -- how fast can MI355X do random 8-byte gathers, and does the load flavour matter?
// Each lane streams int32 indices (16-B nt loads, like the SpMV kernels) and gathers x[idx] (8 B) with one of:
//   0 plain global_load_dwordx2     1 __builtin_nontemporal_load (nt)     2 agent-scope relaxed atomic load (sc1)
//   3 plain, but 4-byte gathers (float table of the same element count)
// Usage: gather_bench <table_MB> <Mgathers>   (hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int int4v __attribute__((ext_vector_type(4)));

template <int MODE> __device__ __forceinline__ double gather(const double *x, int i) {
  if (MODE == 1) return __builtin_nontemporal_load(x + i);
  if (MODE == 2) return __hip_atomic_load(x + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (MODE == 3) return static_cast<double>(reinterpret_cast<const float *>(x)[i]);
  return x[i];
}

template <int MODE> __global__ __launch_bounds__(256) void k(const int *idx, long long n4, const double *x, double *out) {
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

int main(int argc, char **argv) {
  const long long table_mb = argc > 1 ? atoll(argv[1]) : 61;
  const long long n = (argc > 2 ? atoll(argv[2]) : 64) * 1000000LL / 4 * 4;
  const long long elems = table_mb * 1000000LL / 8;
  std::vector<int> h(n);
  unsigned long long st = 88172645463325252ULL;
  for (long long i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = static_cast<int>(st % elems); }
  int *d_idx; double *d_x, *d_out;
  hipMalloc(&d_idx, n * 4); hipMalloc(&d_x, elems * 8); hipMalloc(&d_out, 8);
  hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(d_x, 0, elems * 8);
  const long long n4 = n / 4;
  const int grid = static_cast<int>((n4 + 511) / 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[4] = {"plain dwordx2", "nt dwordx2", "sc1 (agent relaxed) dwordx2", "plain dword (fp32 table)"};
  for (int round = 0; round < 3; ++round)
    for (int mode = 0; mode < 4; ++mode) {
      float best = 1e30f;
      for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d_idx, n4, d_x, d_out);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d_idx, n4, d_x, d_out);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d_idx, n4, d_x, d_out);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d_idx, n4, d_x, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      if (round == 2) printf("table %lld MB, %lld M gathers, %-28s: %8.1f us  %6.1f G gathers/s\n", table_mb, n / 1000000, names[mode], best * 1e3, n / (best * 1e-3) / 1e9);
    }
  return 0;
}
