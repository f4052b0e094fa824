// skewed_gather_bench.hip -- does keeping the HOT part of x in L2 pay when the gathers follow a power law (R-MAT columns)?
// Indices have 25 independent bits, each 1 with probability 0.24 (the column marginal of R-MAT a,b,c,d = .57,.19,.19,.05, scale 25):
// ~60 % of the gathers fall on the 2^19 columns with at most 6 one-bits (4 MB of x = one XCD's L2), the rest spread over 256 MB.
// Each lane streams int32 indices AND 8-byte values with 16-B nt loads (12 B per gather, like SpMV) and gathers x[idx] with:
//   0 plain loads                          1 non-temporal loads for every gather
//   2..5 selective: non-temporal where popcount(idx) > T (T = 5, 6, 7, 8), plain otherwise -- cold gathers then do not displace the hot lines
// Usage: skewed_gather_bench [Mgathers=256]   (hipcc --offload-arch=gfx950 -O3)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int int4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

__global__ void gen(int *idx, double *val, long long n, unsigned long long seed) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    unsigned long long s = seed + i * 0x9E3779B97F4A7C15ULL;
    int c = 0;
    for (int b = 0; b < 25; ++b) {
      s ^= s >> 33; s *= 0xff51afd7ed558ccdULL; s ^= s >> 29; s *= 0xc4ceb9fe1a85ec53ULL; s ^= s >> 32;
      if ((s & 0xffff) < static_cast<unsigned long long>(0.24 * 65536)) c |= 1 << b;
    }
    idx[i] = c;
    val[i] = 1.0;
  }
}

// The selective flavour needs inline assembly: written as `cold ? nontemporal_load(p) : *p` the compiler selects the address and emits ONE
// plain load.  Both loads are issued into the same destination register under complementary exec masks; the caller waits once.
__device__ __forceinline__ void issue_gather(double &r, const double *p, int cold, bool nt_all) {
  unsigned long long save;
  if (nt_all) {
    asm volatile("global_load_dwordx2 %0, %1, off nt" : "=&v"(r) : "v"(p) : "memory");
  } else {
    asm volatile("s_mov_b64 %[save], exec\n\t"
                 "v_cmpx_ne_u32 0, %[cold]\n\t"
                 "global_load_dwordx2 %[r], %[p], off nt\n\t"
                 "s_andn2_b64 exec, %[save], exec\n\t"
                 "global_load_dwordx2 %[r], %[p], off\n\t"
                 "s_mov_b64 exec, %[save]"
                 : [r] "=&v"(r), [save] "=&s"(save)
                 : [p] "v"(p), [cold] "v"(cold)
                 : "vcc", "memory");
  }
}

// The same without hand-written loads: buffer loads, whose cache policy is an immediate operand of the builtin -- two calls that differ in
// it cannot be folded into one, and the compiler does the waiting itself.  (32-bit byte offsets: x below 4 GB.)
template <int AUX> __device__ __forceinline__ double gather_buffer(__amdgpu_buffer_rsrc_t rsrc, int col, bool cold) {
  typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
  uint2v r;
  if (cold) r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, col * 8, 0, AUX); // aux: 1 sc0, 2 nt, 16 sc1 (gfx940 cache-policy bits)
  else r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, col * 8, 0, 0);
  return __hiloint2double(static_cast<int>(r.y), static_cast<int>(r.x));
}

template <int MODE> __global__ __launch_bounds__(256) void k(const int *idx, const double *val, long long n4, const double *x, double *out) {
  const long long t = static_cast<long long>(blockIdx.x) * 512 + threadIdx.x;
  double s = 0;
  int4v a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
  double2v va0 = {0, 0}, va1 = {0, 0}, vb0 = {0, 0}, vb1 = {0, 0};
  const bool fa = t < n4, fb = t + 256 < n4;
  if (fa) {
    a = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(idx) + t);
    va0 = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(val) + 2 * t);
    va1 = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(val) + 2 * t + 1);
  }
  if (fb) {
    b = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(idx) + t + 256);
    vb0 = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(val) + 2 * (t + 256));
    vb1 = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(val) + 2 * (t + 256) + 1);
  }
  if (MODE == 0) {
    if (fa) s += va0.x * x[a.x] + va0.y * x[a.y] + va1.x * x[a.z] + va1.y * x[a.w];
    if (fb) s += vb0.x * x[b.x] + vb0.y * x[b.y] + vb1.x * x[b.z] + vb1.y * x[b.w];
  } else if (MODE >= 6) {
    constexpr int AUX = MODE == 6 ? 2 : MODE == 7 ? 1 : MODE == 8 ? 16 : MODE == 9 ? 17 : MODE == 10 ? 18 : 19;
    const int c[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    double g[8];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = gather_buffer<AUX>(rsrc, c[e], __popc(c[e]) > 6);
    s += va0.x * g[0] + va0.y * g[1] + va1.x * g[2] + va1.y * g[3] + vb0.x * g[4] + vb0.y * g[5] + vb1.x * g[6] + vb1.y * g[7];
  } else {
    // (out-of-range lanes gather x[0]: valid, and weighted by 0)
    const int T = MODE + 3; // modes 2..5: popcount > 5, 6, 7, 8 is cold
    double g[8];
    const int c[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) issue_gather(g[e], x + c[e], __popc(c[e]) > T ? 1 : 0, MODE == 1);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]), "+v"(g[1]), "+v"(g[2]), "+v"(g[3]), "+v"(g[4]), "+v"(g[5]), "+v"(g[6]), "+v"(g[7]) : : "memory");
    s += va0.x * g[0] + va0.y * g[1] + va1.x * g[2] + va1.y * g[3] + vb0.x * g[4] + vb0.y * g[5] + vb1.x * g[6] + vb1.y * g[7];
  }
  if (s == 123.456) out[0] = s; // keep the loads alive
}

int main(int argc, char **argv) {
  const long long n = (argc > 1 ? atoll(argv[1]) : 256) * 1000000LL / 4 * 4;
  const long long elems = 1LL << 25;
  int *d_idx; double *d_val, *d_x, *d_out;
  if (hipMalloc(&d_idx, n * 4) != hipSuccess || hipMalloc(&d_val, n * 8) != hipSuccess || hipMalloc(&d_x, elems * 8) != hipSuccess ||
      hipMalloc(&d_out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipLaunchKernelGGL(gen, dim3(4096), dim3(256), 0, 0, d_idx, d_val, n, 0xC4ULL);
  hipMemset(d_x, 0, elems * 8);
  hipDeviceSynchronize();
  const long long n4 = n / 4;
  const int grid = static_cast<int>((n4 + 511) / 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char *names[12] = {"plain", "nt (all gathers)", "nt where popcount > 5", "nt where popcount > 6", "nt where popcount > 7", "nt where popcount > 8", "popcount > 6, buffer loads nt", "popcount > 6: sc0", "popcount > 6: sc1", "popcount > 6: sc0 sc1", "popcount > 6: nt sc1", "popcount > 6: nt sc0 sc1"};
  for (int round = 0; round < 2; ++round)
    for (int mode = 0; mode < 12; ++mode) {
      float best = 1e30f;
      for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 8) hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 9) hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 10) hipLaunchKernelGGL(k<10>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        if (mode == 11) hipLaunchKernelGGL(k<11>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      if (round == 1) printf("%lld M skewed gathers + 12 B/gather of stream, %-30s: %8.1f us  %6.1f G gathers/s\n", n / 1000000, names[mode], best * 1e3, n / (best * 1e-3) / 1e9);
    }
  return 0;
}
