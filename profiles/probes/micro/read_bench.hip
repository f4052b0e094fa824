// read_bench.hip -- what does MI355X deliver for a READ-ONLY stream (SpMV reads ~12 B per non-zero and writes ~1), against the
// read+write copy the library's `copy_ceiling` probe measures?  Each lane issues K 16-byte loads (nt or plain) per tile of 256*K*16 B.
// Usage: read_bench [MB=2048]   (hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int int4v __attribute__((ext_vector_type(4)));

template <int K, bool NT> __global__ __launch_bounds__(256) void rd(const int4v *__restrict__ p, long long n16, int *out) {
  const long long base = static_cast<long long>(blockIdx.x) * (256 * K) + threadIdx.x;
  int4v v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const long long i = base + k * 256;
    const long long j = i < n16 ? i : 0;
    v[k] = NT ? __builtin_nontemporal_load(p + j) : p[j];
  }
  int s = 0;
#pragma unroll
  for (int k = 0; k < K; ++k) s ^= v[k].x ^ v[k].y ^ v[k].z ^ v[k].w;
  if (s == 0x12345678) out[0] = s;
}

template <int K, bool NT> float run(const int4v *p, long long n16, int *out) {
  const long long blocks = (n16 + 256 * K - 1) / (256 * K);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((rd<K, NT>), dim3(static_cast<unsigned>(blocks)), dim3(256), 0, 0, p, n16, out);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  const long long mb = argc > 1 ? atoll(argv[1]) : 2048;
  const long long bytes = mb << 20, n16 = bytes / 16;
  int4v *p;
  int *out;
  if (hipMalloc(&p, bytes) != hipSuccess || hipMalloc(&out, 4) != hipSuccess) return 1;
  (void)hipMemset(p, 1, bytes);
  (void)hipDeviceSynchronize();
  printf("read-only stream of %lld MB, 16-B loads per lane:\n", mb);
  printf("  K=2  nt %7.1f GB/s   plain %7.1f GB/s\n", bytes / (run<2, true>(p, n16, out) * 1e6), bytes / (run<2, false>(p, n16, out) * 1e6));
  printf("  K=4  nt %7.1f GB/s   plain %7.1f GB/s\n", bytes / (run<4, true>(p, n16, out) * 1e6), bytes / (run<4, false>(p, n16, out) * 1e6));
  printf("  K=6  nt %7.1f GB/s   plain %7.1f GB/s\n", bytes / (run<6, true>(p, n16, out) * 1e6), bytes / (run<6, false>(p, n16, out) * 1e6));
  printf("  K=8  nt %7.1f GB/s   plain %7.1f GB/s\n", bytes / (run<8, true>(p, n16, out) * 1e6), bytes / (run<8, false>(p, n16, out) * 1e6));
  printf("  K=16 nt %7.1f GB/s   plain %7.1f GB/s\n", bytes / (run<16, true>(p, n16, out) * 1e6), bytes / (run<16, false>(p, n16, out) * 1e6));
  return 0;
}
