// xcd_slab_gather_bench.hip -- would routing non-zeros to XCDs by column slab pay on power-law columns (R-MAT)?
// The shipped kernels let every XCD gather from all of x, so each of the eight 4 MB L2s keeps its own copy of the SAME hot lines.
// Here block b only gathers from lines with (line & 7) == (b & 7) -- x lines interleaved over eight slabs, and blocks b, b + 8, ...
// share an XCD (MI355X_MICROARCH.md: blocks are dealt round-robin over the XCDs) -- so the eight L2s hold eight DIFFERENT hot
// sets: 8 x 4 MB of hot lines instead of 4 MB.  Index bits: 25 independent bits, each 1 with probability 0.24 (R-MAT a,b,c,d =
// .57,.19,.19,.05 column marginal); in slab mode bits 4..6 (the line's slab) are overwritten with b & 7.
// Each lane streams int32 indices and 8-byte values with 16-B nt loads (12 B per gather, like SpMV).
//   mode 0: plain gathers, any block gathers anywhere          (the shipped kernel without hints)
//   mode 1: slab-routed indices, plain gathers
//   mode 2: slab-routed, gathers with popcount(other bits) > T non-temporal (hints on top), T = 8
//   mode 3: slab-routed by (b + 1) & 7 ... i.e. the SAME routing shifted by one block: control, must equal mode 1 if placement is by b & 7
//   mode 4: "slab" taken from (blockIdx / 8) & 7: every XCD sees all eight slabs again -- control for the placement assumption
//   round 4 -- how fast could gathers go if EVERY one hit the XCD's L2 (the ceiling of any finer slabbing):
//   mode 5: 64 windows of 4 MB (2^19 columns); block b gathers only inside window 8 * round + (b & 7), round = which eighth of its class's blocks it is in:
//           at any time the eight XCDs work in eight different windows, each window is read by one XCD only
//   mode 6: the same 64 windows walked one after the other by all XCDs together (64 sequential passes: every window enters all eight L2s)
//   mode 7: mode 5 with windows of 2 MB (128 windows, 16 rounds)
// Usage: xcd_slab_gather_bench [Mgathers=512]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int int4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int draw(unsigned long long seed, long long i) {
  unsigned long long s = seed + i * 0x9E3779B97F4A7C15ULL;
  int c = 0;
  for (int b = 0; b < 25; ++b) {
    s ^= s >> 33; s *= 0xff51afd7ed558ccdULL; s ^= s >> 29; s *= 0xc4ceb9fe1a85ec53ULL; s ^= s >> 32;
    if ((s & 0xffff) < static_cast<unsigned long long>(0.24 * 65536)) c |= 1 << b;
  }
  return c;
}

// element i belongs to block i / 2048 of the consuming kernel (512 lanes-steps x 4)
__global__ void gen(int *idx, double *val, long long n, unsigned long long seed, int route) {
  const long long nblk = n / 2048;
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    int c = draw(seed, i);
    const long long blk = i / 2048;
    if (route == 5 || route == 7) {
      const int wbits = route == 5 ? 19 : 18, rounds = route == 5 ? 8 : 16;
      const long long per_class = (nblk + 7) / 8;
      const int round = static_cast<int>(((blk >> 3) * rounds) / per_class);
      c = (c & ((1 << wbits) - 1)) | ((round * 8 + static_cast<int>(blk & 7)) << wbits);
    }
    if (route == 6) c = (c & ((1 << 19) - 1)) | (static_cast<int>((blk * 64) / nblk) << 19);
    if (route == 1) c = (c & ~0x70) | (static_cast<int>(blk & 7) << 4);
    if (route == 3) c = (c & ~0x70) | (static_cast<int>((blk + 1) & 7) << 4);
    if (route == 4) c = (c & ~0x70) | (static_cast<int>((blk >> 3) & 7) << 4);
    idx[i] = c;
    val[i] = 1.0;
  }
}

template <int AUX> __device__ __forceinline__ double gather_buffer(__amdgpu_buffer_rsrc_t rsrc, int col, bool cold) {
  typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
  uint2v r;
  if (cold) r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, col * 8, 0, AUX);
  else r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, col * 8, 0, 0);
  return __hiloint2double(static_cast<int>(r.y), static_cast<int>(r.x));
}

template <int HINT_T> __global__ __launch_bounds__(256) void k(const int *idx, const double *val, long long n4, const double *x, double *out) {
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
  if (HINT_T == 0) {
    if (fa) s += va0.x * x[a.x] + va0.y * x[a.y] + va1.x * x[a.z] + va1.y * x[a.w];
    if (fb) s += vb0.x * x[b.x] + vb0.y * x[b.y] + vb1.x * x[b.z] + vb1.y * x[b.w];
  } else {
    const int c[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    double g[8];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int e = 0; e < 8; ++e) g[e] = gather_buffer<2>(rsrc, c[e], __popc(c[e] & ~0x70) > HINT_T);
    s += va0.x * g[0] + va0.y * g[1] + va1.x * g[2] + va1.y * g[3] + vb0.x * g[4] + vb0.y * g[5] + vb1.x * g[6] + vb1.y * g[7];
  }
  if (s == 123.456) out[0] = s;
}

int main(int argc, char **argv) {
  const long long n = (argc > 1 ? atoll(argv[1]) : 512) * 1000000LL / 2048 * 2048;
  const long long elems = 1LL << 25;
  int *d_idx; double *d_val, *d_x, *d_out;
  if (hipMalloc(&d_idx, n * 4) != hipSuccess || hipMalloc(&d_val, n * 8) != hipSuccess || hipMalloc(&d_x, elems * 8) != hipSuccess ||
      hipMalloc(&d_out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(d_x, 0, elems * 8);
  const long long n4 = n / 4;
  const int grid = static_cast<int>((n4 + 511) / 512);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  struct { const char *name; int route; int hint; } modes[] = {
      {"any block gathers anywhere, plain (shipped, no hints)", 0, 0},
      {"any block gathers anywhere, popcount > 6 non-temporal (shipped hints)", 0, 6},
      {"slab-routed: block b gathers lines with (line & 7) == (b & 7), plain", 1, 0},
      {"slab-routed, popcount(other bits) > 8 non-temporal", 1, 8},
      {"slab-routed, popcount(other bits) > 9 non-temporal", 1, 9},
      {"slab-routed, popcount(other bits) > 10 non-temporal", 1, 10},
      {"control: slab (b + 1) & 7 (same placement, shifted)", 3, 0},
      {"control: slab (b / 8) & 7 (every XCD sees every slab)", 4, 0},
      {"64 windows of 4 MB, eight at a time, one per XCD (all gathers L2 hits)", 5, 0},
      {"64 windows of 4 MB one after the other, all XCDs in the same window", 6, 0},
      {"128 windows of 2 MB, eight at a time, one per XCD", 7, 0},
  };
  for (auto &m : modes) {
    hipLaunchKernelGGL(gen, dim3(8192), dim3(256), 0, 0, d_idx, d_val, n, 0xC4ULL, m.route);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(e0);
      if (m.hint == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
      if (m.hint == 6) hipLaunchKernelGGL(k<6>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
      if (m.hint == 8) hipLaunchKernelGGL(k<8>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
      if (m.hint == 9) hipLaunchKernelGGL(k<9>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
      if (m.hint == 10) hipLaunchKernelGGL(k<10>, dim3(grid), dim3(256), 0, 0, d_idx, d_val, n4, d_x, d_out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
    printf("%lld M gathers + 12 B/gather of nt stream | %-72s: %8.1f us  %6.1f G gathers/s\n", n / 1000000, m.name, best * 1e3, n / (best * 1e-3) / 1e9);
  }
  return 0;
}
