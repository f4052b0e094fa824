// scalar_gather_request_size_bench.hip -- round 6: does a SCALAR load (s_load_dwordx2: scalar cache, 64-B lines) that misses the L2 move fewer bytes over the
// fabric than a vector load's 128-B line?  (gather_request_size_bench.hip, round 5: every vector flavour is one 128-B request per random 8-byte gather;
// a few hundred 64-B requests per dispatch were left unexplained.)  Each wavefront reads 64 indices (one per lane), then gathers them ONE PER s_load
// from a table far beyond the caches: v_readlane -> s_load_dwordx2 (SGPR offset) -> sum.  Eight loads in flight per wavefront.
// Run under `rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum`.
// Usage: scalar_gather_request_size_bench <table_MB> <Mgathers>       (hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int INFLIGHT> __global__ __launch_bounds__(256) void scalar_gathers(const int *idx, long long n, const double *x, double *out) {
  const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const int mine = t < n ? idx[t] : 0;
  double s = 0.0;
#pragma unroll 1
  for (int l = 0; l < 64; l += INFLIGHT) {
    double v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) {
      const unsigned off = static_cast<unsigned>(__builtin_amdgcn_readlane(mine, l + k)) << 3;
      asm volatile("s_load_dwordx2 %0, %1, %2" : "=s"(v[k]) : "s"(x), "s"(off) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < INFLIGHT; ++k) s += v[k];
  }
  if (s == 123.456) out[0] = s;
}

__global__ __launch_bounds__(256) void vector_gathers(const int *idx, long long n, const double *x, double *out) {
  const long long t = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  const int mine = t < n ? idx[t] : 0;
  const double s = x[mine];
  if (s == 123.456) out[0] = s;
}

int main(int argc, char **argv) {
  const long long table_mb = argc > 1 ? atoll(argv[1]) : 512;
  const long long n = (argc > 2 ? atoll(argv[2]) : 16) * 1000000LL / 256 * 256;
  const long long elems = table_mb * 1000000LL / 8;
  std::vector<int> h(n);
  unsigned long long st = 88172645463325252ULL;
  for (long long i = 0; i < n; ++i) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; h[i] = static_cast<int>(st % elems); }
  int *d_idx; double *d_out, *tab;
  hipMalloc(&d_idx, n * 4); hipMalloc(&d_out, 8); hipMalloc(&tab, elems * 8);
  hipMemcpy(d_idx, h.data(), n * 4, hipMemcpyHostToDevice);
  hipMemset(tab, 0, elems * 8);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid = static_cast<int>(n / 256);
  for (int which = 0; which < 4; ++which) {
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(scalar_gathers<8>, dim3(grid), dim3(256), 0, 0, d_idx, n, tab, d_out);
      else if (which == 1) hipLaunchKernelGGL(scalar_gathers<16>, dim3(grid), dim3(256), 0, 0, d_idx, n, tab, d_out);
      else if (which == 2) hipLaunchKernelGGL(scalar_gathers<32>, dim3(grid), dim3(256), 0, 0, d_idx, n, tab, d_out);
      else hipLaunchKernelGGL(vector_gathers, dim3(grid), dim3(256), 0, 0, d_idx, n, tab, d_out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    const char *names[4] = {"s_load_dwordx2, 8 in flight per wavefront", "s_load_dwordx2, 16 in flight", "s_load_dwordx2, 32 in flight", "global_load_dwordx2 (one per lane)"};
    printf("table %lld MB, %lld M gathers, %s: %.1f us  %.2f G gathers/s\n", table_mb, n / 1000000, names[which], best * 1e3, n / (best * 1e-3) / 1e9);
  }
  return 0;
}
