// read_write_mix_bench.hip -- round 5: what do the y stores of a short-row SpMV cost the memory system?  (profiles/r05_short_row_dissection.txt: on the
// banded shard of BASELINE configs[4] removing the y store -- 7 % of the kernel's bytes -- removes 20 % of its time.)  A streaming kernel in the tile
// kernels' shape: every 256-thread workgroup reads 24 KB (16-B loads, six per lane, default policy or nt) and writes W bytes of a second array in one of
// several forms.  Prints GB/s of READ bytes for each form; the read-only row is the yardstick.
//   store form: 0 none | 1 8 B per lane, 256 lanes (2 KB, the SpMV's y store) | 2 16 B per lane, 128 lanes (2 KB) | 3 form 1 non-temporal
//               | 4 8 B per lane, 64 lanes (512 B: a quarter of the rows) | 5 form 1 to addresses aligned to 128 B per wavefront (already the case here)
// Usage: read_write_mix_bench <read_MB> [nt_loads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int int4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

template <int FORM, bool NT> __global__ __launch_bounds__(256) void k(const int4v *__restrict__ src, long long n16, double *__restrict__ dst) {
  const long long base = static_cast<long long>(blockIdx.x) * (256 * 6);
  int4v acc = {0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const long long j = base + i * 256 + threadIdx.x;
    if (j < n16) {
      const int4v v = NT ? __builtin_nontemporal_load(src + j) : src[j];
      acc += v;
    }
  }
  const double r = static_cast<double>(acc.x + acc.y + acc.z + acc.w);
  const long long row = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x;
  if (FORM == 1 || FORM == 5) dst[row] = r;
  if (FORM == 3) __builtin_nontemporal_store(r, dst + row);
  if (FORM == 2) {
    const double other = __shfl_down(r, 1, 64);
    if ((threadIdx.x & 1) == 0) {
      double2v p = {r, other};
      *reinterpret_cast<double2v *>(dst + row) = p;
    }
  }
  if (FORM == 4 && threadIdx.x < 64) dst[static_cast<long long>(blockIdx.x) * 64 + threadIdx.x] = r;
  if (FORM == 0 && r == 123.456) dst[0] = r;
}

template <int FORM, bool NT> float run(int grid, const int4v *src, long long n16, double *dst, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<FORM, NT>), dim3(grid), dim3(256), 0, 0, src, n16, dst);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  const long long read_mb = argc > 1 ? atoll(argv[1]) : 3072;
  const bool nt = argc > 2 && atoi(argv[2]) == 1;
  const long long n16 = read_mb * 1000000LL / 16;
  const int grid = static_cast<int>((n16 + 256 * 6 - 1) / (256 * 6));
  int4v *src;
  double *dst;
  (void)hipMalloc(&src, n16 * 16);
  (void)hipMalloc(&dst, static_cast<size_t>(grid) * 256 * 8 + 64);
  (void)hipMemset(src, 1, n16 * 16);
  (void)hipMemset(dst, 0, static_cast<size_t>(grid) * 256 * 8);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const char *names[6] = {"no store", "8 B x 256 lanes (2 KB per workgroup)", "16 B x 128 lanes (2 KB)", "8 B x 256 lanes, non-temporal", "8 B x 64 lanes (512 B)", "-"};
  float ms[5];
  if (nt) {
    ms[0] = run<0, true>(grid, src, n16, dst, e0, e1); ms[1] = run<1, true>(grid, src, n16, dst, e0, e1); ms[2] = run<2, true>(grid, src, n16, dst, e0, e1);
    ms[3] = run<3, true>(grid, src, n16, dst, e0, e1); ms[4] = run<4, true>(grid, src, n16, dst, e0, e1);
  } else {
    ms[0] = run<0, false>(grid, src, n16, dst, e0, e1); ms[1] = run<1, false>(grid, src, n16, dst, e0, e1); ms[2] = run<2, false>(grid, src, n16, dst, e0, e1);
    ms[3] = run<3, false>(grid, src, n16, dst, e0, e1); ms[4] = run<4, false>(grid, src, n16, dst, e0, e1);
  }
  for (int f = 0; f < 5; ++f)
    printf("read %lld MB (%s loads), 24 KB per workgroup, store form %d %-38s: %8.1f us  %7.1f GB/s of read bytes  (x %.3f of read-only)\n", read_mb,
           nt ? "nt" : "default", f, names[f], ms[f] * 1e3, read_mb * 1e6 / (ms[f] * 1e-3) / 1e9, ms[f] / ms[0]);
  return 0;
}
