// spmv_skeleton_bench.hip -- why does a tile kernel stream at 5.3-6.6 TB/s when a bare read loop reaches 7.2 (read_bench.hip)?
// The skeleton keeps the life cycle of the library's tile kernels and nothing else: per 2048-"non-zero" tile a workgroup streams 8 KB of
// int32 indices + 16 KB of fp64 values (16-B loads, nt or plain), gathers x[index] from a 3 MB table (L2 hits, like a FEM matrix' x),
// multiplies into LDS, barrier, every lane sums 8 products, one 8-byte store per lane group of 16.
//   A: one tile per workgroup (the library's form): the stream loads of a workgroup are in flight for a fraction of its life only
//   B: persistent workgroups, the NEXT tile's stream loads issued before the current tile's gather / LDS / sum phase (software pipeline)
// Usage: spmv_skeleton_bench [Mnnz=64]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef int int4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));
constexpr int kT = 256, kTile = 2048;

template <bool NT> __device__ __forceinline__ int4v ld_i4(const int *p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const int4v *>(p)) : *reinterpret_cast<const int4v *>(p);
}
template <bool NT> __device__ __forceinline__ double2v ld_d2(const double *p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const double2v *>(p)) : *reinterpret_cast<const double2v *>(p);
}

struct Regs {
  int4v c[2];
  double2v a[2], b[2];
};
template <bool NT> __device__ __forceinline__ void issue(Regs &r, const int *ci, const double *v, long long t) {
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long j = t * kTile + 4 * (threadIdx.x + k * kT);
    r.c[k] = ld_i4<NT>(ci + j);
    r.a[k] = ld_d2<NT>(v + j);
    r.b[k] = ld_d2<NT>(v + j + 2);
  }
}
__device__ __forceinline__ void finish(const Regs &r, const double *x, double *lds) {
  double g[2][4];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    g[k][0] = x[r.c[k].x];
    g[k][1] = x[r.c[k].y];
    g[k][2] = x[r.c[k].z];
    g[k][3] = x[r.c[k].w];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double2v p0, p1;
    p0.x = r.a[k].x * g[k][0];
    p0.y = r.a[k].y * g[k][1];
    p1.x = r.b[k].x * g[k][2];
    p1.y = r.b[k].y * g[k][3];
    double2v *dst = reinterpret_cast<double2v *>(lds + 4 * (threadIdx.x + k * kT));
    dst[0] = p0;
    dst[1] = p1;
  }
}
__device__ __forceinline__ void rows(const double *lds, double *y, long long t) {
  double s = 0;
#pragma unroll
  for (int e = 0; e < 8; ++e) s += lds[threadIdx.x + e * kT];
  // 16 lanes per "row" of 128 non-zeros
  s += __shfl_xor(s, 1);
  s += __shfl_xor(s, 2);
  s += __shfl_xor(s, 4);
  s += __shfl_xor(s, 8);
  if ((threadIdx.x & 15) == 0) y[t * (kT / 16) + threadIdx.x / 16] = s;
}

template <bool NT> __global__ __launch_bounds__(kT) void one_tile(const int *ci, const double *v, const double *x, double *y, long long ntiles) {
  __shared__ __attribute__((aligned(16))) double lds[kTile];
  const long long t = blockIdx.x;
  Regs r;
  issue<NT>(r, ci, v, t);
  finish(r, x, lds);
  __syncthreads();
  rows(lds, y, t);
}

// Transposed gathers: the stream loads stay 16 B per lane (4 consecutive non-zeros), but the column ids cross the wave through LDS so that gather
// instruction e serves non-zeros e*64 + lane of the wave's 256: consecutive non-zeros -> (on run-structured matrices) consecutive columns -> 4-8 cache
// lines per instruction instead of 16-32.  The x values land transposed in the tile slots, each lane reads its four back and multiplies in place.
template <bool NT> __global__ __launch_bounds__(kT) void one_tile_t(const int *ci, const double *v, const double *x, double *y, long long ntiles) {
  __shared__ __attribute__((aligned(16))) double lds[kTile];
  __shared__ __attribute__((aligned(16))) int lci[kTile];
  const long long t = blockIdx.x;
  Regs r;
  issue<NT>(r, ci, v, t);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int *cw = lci + (k * 4 + wave) * 256;
    double *xw = lds + (k * 4 + wave) * 256;
    *reinterpret_cast<int4v *>(cw + 4 * lane) = r.c[k];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    int ct[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) ct[e] = cw[e * 64 + lane];
    double xt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) xt[e] = x[ct[e]];
#pragma unroll
    for (int e = 0; e < 4; ++e) xw[e * 64 + lane] = xt[e];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double2v x01 = *reinterpret_cast<double2v *>(xw + 4 * lane), x23 = *reinterpret_cast<double2v *>(xw + 4 * lane + 2);
    double2v p0, p1;
    p0.x = r.a[k].x * x01.x;
    p0.y = r.a[k].y * x01.y;
    p1.x = r.b[k].x * x23.x;
    p1.y = r.b[k].y * x23.y;
    *reinterpret_cast<double2v *>(xw + 4 * lane) = p0;
    *reinterpret_cast<double2v *>(xw + 4 * lane + 2) = p1;
  }
  __syncthreads();
  rows(lds, y, t);
}

// Pair mapping: every load instruction of the wave covers whole cache lines.  Per 256-non-zero wave chunk lane l owns the pairs {2l, 2l+1} and
// {128+2l, 129+2l}: values as two 16-B loads from two contiguous 1-KB segments, columns as two 8-B loads from two contiguous 512-B segments
// (the 4-consecutive mapping reads values as two 16-B loads at a 32-B lane stride: each instruction touches every line of a 2-KB span and uses half).
typedef int int2v __attribute__((ext_vector_type(2)));
template <bool NT> __global__ __launch_bounds__(kT) void one_tile_pairs(const int *ci, const double *v, const double *x, double *y, long long ntiles) {
  __shared__ __attribute__((aligned(16))) double lds[kTile];
  const long long t = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int2v c[2][2];
  double2v a[2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const long long chunk = t * kTile + (k * 4 + wave) * 256;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long long j = chunk + h * 128 + 2 * lane;
      c[k][h] = NT ? __builtin_nontemporal_load(reinterpret_cast<const int2v *>(ci + j)) : *reinterpret_cast<const int2v *>(ci + j);
      a[k][h] = ld_d2<NT>(v + j);
    }
  }
  double g[2][2][2];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      g[k][h][0] = x[c[k][h].x];
      g[k][h][1] = x[c[k][h].y];
    }
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      double2v p;
      p.x = a[k][h].x * g[k][h][0];
      p.y = a[k][h].y * g[k][h][1];
      *reinterpret_cast<double2v *>(lds + (k * 4 + wave) * 256 + h * 128 + 2 * lane) = p;
    }
  __syncthreads();
  rows(lds, y, t);
}

// dissection of the one-tile form: MODE bit 0 = gathers, bit 1 = LDS tile + barrier (else the products are summed in registers)
template <bool NT, int MODE> __global__ __launch_bounds__(kT) void dissect(const int *ci, const double *v, const double *x, double *y, long long ntiles) {
  __shared__ __attribute__((aligned(16))) double lds[kTile];
  const long long t = blockIdx.x;
  Regs r;
  issue<NT>(r, ci, v, t);
  double s = 0;
  if (MODE & 2) {
    if (MODE & 1) finish(r, x, lds);
    else {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        double2v p0, p1;
        p0.x = r.a[k].x * r.c[k].x; p0.y = r.a[k].y * r.c[k].y; p1.x = r.b[k].x * r.c[k].z; p1.y = r.b[k].y * r.c[k].w;
        double2v *dst = reinterpret_cast<double2v *>(lds + 4 * (threadIdx.x + k * kT));
        dst[0] = p0; dst[1] = p1;
      }
    }
    __syncthreads();
    rows(lds, y, t);
  } else {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (MODE & 1) s += r.a[k].x * x[r.c[k].x] + r.a[k].y * x[r.c[k].y] + r.b[k].x * x[r.c[k].z] + r.b[k].y * x[r.c[k].w];
      else s += r.a[k].x * r.c[k].x + r.a[k].y * r.c[k].y + r.b[k].x * r.c[k].z + r.b[k].y * r.c[k].w;
    }
    s += __shfl_xor(s, 1); s += __shfl_xor(s, 2); s += __shfl_xor(s, 4); s += __shfl_xor(s, 8);
    if ((threadIdx.x & 15) == 0) y[t * (kT / 16) + threadIdx.x / 16] = s;
  }
}

template <bool NT> __global__ __launch_bounds__(kT) void pipelined(const int *ci, const double *v, const double *x, double *y, long long ntiles) {
  __shared__ __attribute__((aligned(16))) double lds[kTile];
  long long t = blockIdx.x;
  if (t >= ntiles) return;
  Regs r;
  issue<NT>(r, ci, v, t);
  while (true) {
    finish(r, x, lds);
    const long long tn = t + gridDim.x;
    const bool more = tn < ntiles; // workgroup-uniform
    if (more) issue<NT>(r, ci, v, tn); // the next tile's stream is in flight during the barrier, the sums and the store
    __syncthreads();
    rows(lds, y, t);
    if (!more) break;
    __syncthreads(); // the tile is overwritten next
    t = tn;
  }
}

// persistent without the prefetch: separates "fewer workgroup launches" from "stream loads in flight all the time"
template <bool NT> __global__ __launch_bounds__(kT) void persistent_plain(const int *ci, const double *v, const double *x, double *y, long long ntiles) {
  __shared__ __attribute__((aligned(16))) double lds[kTile];
  for (long long t = blockIdx.x; t < ntiles; t += gridDim.x) {
    Regs r;
    issue<NT>(r, ci, v, t);
    finish(r, x, lds);
    __syncthreads();
    rows(lds, y, t);
    __syncthreads();
  }
}

__global__ void fill(int *ci, double *v, long long n, int xn, int pattern) {
  for (long long i = blockIdx.x * 256LL + threadIdx.x; i < n; i += gridDim.x * 256LL) {
    if (pattern == 0) ci[i] = static_cast<int>((i / 3 + (i % 3) * 1000) % xn); // three interleaved clusters, no runs
    else {
      // rows of 98 non-zeros, a run of consecutive (pattern 1) or every-other (pattern 2) columns around the diagonal: the FEM-like stand-ins
      const long long row = i / 98, k = i % 98;
      ci[i] = static_cast<int>((row * xn / (n / 98 + 1) + k * pattern) % xn);
    }
    v[i] = 1.0;
  }
}

template <class K> float time_it(K launch) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int r = 0; r < 6; ++r) {
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

int main(int argc, char **argv) {
  const long long nnz = (argc > 1 ? atoll(argv[1]) : 64) * 1000000LL / kTile * kTile, ntiles = nnz / kTile;
  const int xn = 381689;
  int *ci;
  double *v, *x, *y;
  if (hipMalloc(&ci, nnz * 4) != hipSuccess || hipMalloc(&v, nnz * 8) != hipSuccess || hipMalloc(&x, xn * 8) != hipSuccess ||
      hipMalloc(&y, ntiles * (kT / 16) * 8) != hipSuccess)
    return 1;
  const int pattern = argc > 2 ? atoi(argv[2]) : 0;
  hipLaunchKernelGGL(fill, dim3(4096), dim3(256), 0, 0, ci, v, nnz, xn, pattern);
  (void)hipMemset(x, 0, xn * 8);
  (void)hipDeviceSynchronize();
  const double gb = nnz * 12.0 / 1e9;
  printf("%lld M non-zeros (%.0f MB of stream), x = 3 MB, column pattern %d:\n", nnz / 1000000, gb * 1e3, pattern);
  const unsigned g1 = static_cast<unsigned>(ntiles);
  printf("  one tile per workgroup            nt %7.1f GB/s   plain %7.1f GB/s\n",
         gb / (time_it([&] { hipLaunchKernelGGL(one_tile<true>, dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL(one_tile<false>, dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3));
  printf("  one tile, transposed gathers      nt %7.1f GB/s   plain %7.1f GB/s\n",
         gb / (time_it([&] { hipLaunchKernelGGL(one_tile_t<true>, dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL(one_tile_t<false>, dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3));
  printf("  one tile, pair mapping            nt %7.1f GB/s   plain %7.1f GB/s\n",
         gb / (time_it([&] { hipLaunchKernelGGL(one_tile_pairs<true>, dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL(one_tile_pairs<false>, dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3));
  printf("  dissection (one tile per workgroup, nt): no gathers no LDS %7.1f | gathers only %7.1f | LDS+barrier only %7.1f | both %7.1f GB/s\n",
         gb / (time_it([&] { hipLaunchKernelGGL((dissect<true, 0>), dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL((dissect<true, 1>), dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL((dissect<true, 2>), dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL((dissect<true, 3>), dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3));
  printf("  dissection (one tile per workgroup, plain): no gathers no LDS %7.1f | gathers only %7.1f | LDS+barrier only %7.1f | both %7.1f GB/s\n",
         gb / (time_it([&] { hipLaunchKernelGGL((dissect<false, 0>), dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL((dissect<false, 1>), dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL((dissect<false, 2>), dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
         gb / (time_it([&] { hipLaunchKernelGGL((dissect<false, 3>), dim3(g1), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3));
  for (int per_cu : {8}) {
    const unsigned g = 256u * per_cu;
    printf("  persistent, %d workgroups per CU   nt %7.1f GB/s   plain %7.1f GB/s\n", per_cu,
           gb / (time_it([&] { hipLaunchKernelGGL(persistent_plain<true>, dim3(g), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
           gb / (time_it([&] { hipLaunchKernelGGL(persistent_plain<false>, dim3(g), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3));
    printf("  pipelined,  %d workgroups per CU   nt %7.1f GB/s   plain %7.1f GB/s\n", per_cu,
           gb / (time_it([&] { hipLaunchKernelGGL(pipelined<true>, dim3(g), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3),
           gb / (time_it([&] { hipLaunchKernelGGL(pipelined<false>, dim3(g), dim3(kT), 0, 0, ci, v, x, y, ntiles); }) * 1e-3));
  }
  return 0;
}
