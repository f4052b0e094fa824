// far_split_bench.hip -- round 4, VERDICT item 2: what would an XCD-affine column-slab pass over the FAR non-zeros of the headline stand-in cost?
// Stand-in parameters (synth.hardesty3_like_torch): n = 7,591,564 columns (x = 60.7 MB), 40,451,632 non-zeros of which 10 % (4,045,163) have
// uniformly random columns; today each of those costs the row-block kernel one 64-B fabric sector (PMC: 954 MB moved for 710.5 MB useful; far-free variant
// of the same matrix 112 us against 155-158 us with them, i.e. ~43-46 us for the far gathers).
// The design under test: a FAR PASS ahead of the SpMV gathers xg[k] = x[farcol[k]] for the plan's list of far columns, sorted by column slab, slab s
// handled only by workgroups with blockIdx % 8 == s % 8 (one XCD's L2 then holds the slab: every x line enters ONE L2 once); the SpMV kernel would read xg
// as a stream (8 B per far non-zero) plus 2-4 B of index per far non-zero instead of gathering.
//   mode 0: far columns in matrix order (random), every block anywhere                      -- what the gathers cost as a separate pass, no locality
//   mode 1: sorted by slab (S slabs of n / S columns), blocks walk the list in order       -- every XCD sees every slab (8 L2s replicate it)
//   mode 2: sorted by slab, slab s only on workgroups with b % 8 == s % 8 (XCD-affine)     -- the proposal
//   mode 3: mode 2 + each slab's part of x prefetched by its XCD first (one 16-B load per lane over the slab)
// Each mode: reads 4 B of column per element (nt), gathers 8 B, writes 8 B (nt) to xg in list order.  Also timed: the consumer's extra stream (12 B per far
// element read) as a plain copy-rate figure.
// Usage: far_split_bench [n=7591564] [far=4045163] [S=16]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

__global__ __launch_bounds__(256) void far_pass_plain(const int *__restrict__ col, long long cnt, const double *__restrict__ x, double *__restrict__ xg) {
  const long long i0 = (static_cast<long long>(blockIdx.x) * 256 + threadIdx.x) * 4;
  if (i0 + 4 <= cnt) {
    typedef int int4v __attribute__((ext_vector_type(4)));
    typedef double double2v __attribute__((ext_vector_type(2)));
    const int4v c = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(col + i0));
    const double a = x[c.x], b = x[c.y], d = x[c.z], e = x[c.w];
    __builtin_nontemporal_store(double2v{a, b}, reinterpret_cast<double2v *>(xg + i0));
    __builtin_nontemporal_store(double2v{d, e}, reinterpret_cast<double2v *>(xg + i0 + 2));
  } else {
    for (long long i = i0; i < cnt; ++i) xg[i] = x[col[i]];
  }
}

// XCD-affine: class c = blockIdx % 8 owns slabs c, c + 8, ...; cls_first[c * (R + 1) + r] = first list element of the class's r-th slab (padded to 4),
// cls_blk[c * (R + 1) + r] = first block (within the class) of that slab.  A block finds its slab by a scan over R <= 8 entries.
template <bool PREFETCH>
__global__ __launch_bounds__(256) void far_pass_affine(const int *__restrict__ col, const long long *__restrict__ slab_first, const int *__restrict__ cls_blk,
                                                       int R, int S, int slab_cols, int n, const double *__restrict__ x, double *__restrict__ xg) {
  const int c = blockIdx.x & 7, i = blockIdx.x >> 3;
  int r = 0;
  while (r + 1 < R && i >= cls_blk[c * (R + 1) + r + 1]) ++r;
  if (i >= cls_blk[c * (R + 1) + R]) return;
  const int s = c + 8 * r;
  if (s >= S) return;
  const long long lo = slab_first[s], hi = slab_first[s + 1];
  const int bi = i - cls_blk[c * (R + 1) + r];
  if (PREFETCH) { // the first blocks of a slab pull the slab's x range into this XCD's L2 (16 B per lane, 4 KB per block step)
    const int nb = cls_blk[c * (R + 1) + r + 1] - cls_blk[c * (R + 1) + r];
    const long long c0 = static_cast<long long>(s) * slab_cols, c1 = c0 + slab_cols < n ? c0 + slab_cols : n;
    double sink = 0;
    for (long long j = c0 + (static_cast<long long>(bi) * 256 + threadIdx.x) * 16; j < c1; j += static_cast<long long>(nb) * 256 * 16) sink += x[j];
    if (sink == 123.456) xg[0] = sink;
  }
  const long long i0 = lo + (static_cast<long long>(bi) * 256 + threadIdx.x) * 4;
  if (i0 >= hi) return;
  if (i0 + 4 <= hi) {
    typedef int int4v __attribute__((ext_vector_type(4)));
    typedef double double2v __attribute__((ext_vector_type(2)));
    const int4v cc = __builtin_nontemporal_load(reinterpret_cast<const int4v *>(col + i0));
    const double a = x[cc.x], b = x[cc.y], d = x[cc.z], e = x[cc.w];
    __builtin_nontemporal_store(double2v{a, b}, reinterpret_cast<double2v *>(xg + i0));
    __builtin_nontemporal_store(double2v{d, e}, reinterpret_cast<double2v *>(xg + i0 + 2));
  } else {
    for (long long k = i0; k < hi; ++k) xg[k] = x[col[k]];
  }
}

__global__ __launch_bounds__(256) void stream_read(const double *__restrict__ a, long long n16, double *out) {
  typedef double double2v __attribute__((ext_vector_type(2)));
  double s = 0;
  for (long long i = static_cast<long long>(blockIdx.x) * 256 + threadIdx.x; i < n16; i += static_cast<long long>(gridDim.x) * 256) {
    const double2v v = __builtin_nontemporal_load(reinterpret_cast<const double2v *>(a) + i);
    s += v.x + v.y;
  }
  if (s == 123.456) out[0] = s;
}

static float time_it(int reps, hipEvent_t e0, hipEvent_t e1, void *flush, size_t flush_bytes, const std::function<void()> &f) {
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipMemsetAsync(flush, r, flush_bytes, 0); // 512 MB written: x and the lists leave L2 and the Infinity Cache, as between two SpMVs of a 710 MB matrix
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r > 0 && ms < best) best = ms;
  }
  return best;
}

#include <functional>
int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int n = argc > 1 ? atoi(argv[1]) : 7591564;
  const long long far = argc > 2 ? atoll(argv[2]) : 4045163;
  const int S = argc > 3 ? atoi(argv[3]) : 16;
  const int slab_cols = (n + S - 1) / S;
  std::mt19937_64 rng(0xC2);
  std::vector<int> col(far);
  for (auto &c : col) c = static_cast<int>(rng() % static_cast<unsigned long long>(n));
  // sorted by slab, matrix order kept inside a slab (stable); every slab's part padded to a multiple of 4 with its own first column
  std::vector<std::vector<int>> by(S);
  for (int c : col) by[c / slab_cols].push_back(c);
  std::vector<int> sorted;
  std::vector<long long> slab_first(S + 1, 0);
  for (int s = 0; s < S; ++s) {
    slab_first[s] = static_cast<long long>(sorted.size());
    sorted.insert(sorted.end(), by[s].begin(), by[s].end());
    while (sorted.size() % 4) sorted.push_back(by[s].empty() ? 0 : by[s][0]);
  }
  slab_first[S] = static_cast<long long>(sorted.size());
  const int R = (S + 7) / 8;
  std::vector<int> cls_blk(8 * (R + 1), 0);
  int max_cls = 0;
  for (int c = 0; c < 8; ++c) {
    int at = 0;
    for (int r = 0; r < R; ++r) {
      cls_blk[c * (R + 1) + r] = at;
      const int s = c + 8 * r;
      if (s < S) at += static_cast<int>((slab_first[s + 1] - slab_first[s] + 1023) / 1024);
    }
    cls_blk[c * (R + 1) + R] = at;
    max_cls = std::max(max_cls, at);
  }
  int *d_col, *d_sorted, *d_cls;
  long long *d_first;
  double *d_x, *d_xg, *d_out;
  void *d_flush;
  const size_t flush_bytes = 512u << 20;
  const long long padded = static_cast<long long>(sorted.size());
  hipMalloc(&d_col, (far + 4) * 4); hipMalloc(&d_sorted, padded * 4); hipMalloc(&d_cls, cls_blk.size() * 4); hipMalloc(&d_first, (S + 1) * 8);
  hipMalloc(&d_x, static_cast<size_t>(n) * 8); hipMalloc(&d_xg, (padded + 4) * 8 + static_cast<size_t>(far) * 4 + 64); /* + room for the consumer-side read of 12 B per element */ hipMalloc(&d_out, 8); hipMalloc(&d_flush, flush_bytes);
  hipMemcpy(d_col, col.data(), far * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_sorted, sorted.data(), padded * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_cls, cls_blk.data(), cls_blk.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_first, slab_first.data(), (S + 1) * 8, hipMemcpyHostToDevice);
  hipMemset(d_x, 0, static_cast<size_t>(n) * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int grid_plain = static_cast<int>((far + 1023) / 1024), grid_sorted = static_cast<int>((padded + 1023) / 1024);
  printf("n %d (x %.1f MB), %lld far gathers, %d slabs of %d columns (%.2f MB of x each), %d slabs per XCD class\n", n, n * 8e-6, far, S, slab_cols,
         slab_cols * 8e-6, R);
  float t;
  t = time_it(6, e0, e1, d_flush, flush_bytes, [&] { hipLaunchKernelGGL(far_pass_plain, dim3(grid_plain), dim3(256), 0, 0, d_col, far, d_x, d_xg); });
  printf("mode 0  matrix order, no locality                         : %7.1f us  %6.1f G gathers/s\n", t * 1e3, far / (t * 1e-3) / 1e9);
  t = time_it(6, e0, e1, d_flush, flush_bytes, [&] { hipLaunchKernelGGL(far_pass_plain, dim3(grid_sorted), dim3(256), 0, 0, d_sorted, padded, d_x, d_xg); });
  printf("mode 1  sorted by slab, blocks in list order (all XCDs)    : %7.1f us  %6.1f G gathers/s\n", t * 1e3, far / (t * 1e-3) / 1e9);
  t = time_it(6, e0, e1, d_flush, flush_bytes, [&] {
    hipLaunchKernelGGL(far_pass_affine<false>, dim3(8 * max_cls), dim3(256), 0, 0, d_sorted, d_first, d_cls, R, S, slab_cols, n, d_x, d_xg);
  });
  printf("mode 2  sorted by slab, slab s on blocks b %% 8 == s %% 8     : %7.1f us  %6.1f G gathers/s\n", t * 1e3, far / (t * 1e-3) / 1e9);
  t = time_it(6, e0, e1, d_flush, flush_bytes, [&] {
    hipLaunchKernelGGL(far_pass_affine<true>, dim3(8 * max_cls), dim3(256), 0, 0, d_sorted, d_first, d_cls, R, S, slab_cols, n, d_x, d_xg);
  });
  printf("mode 3  mode 2 + the slab's x prefetched by its XCD        : %7.1f us  %6.1f G gathers/s\n", t * 1e3, far / (t * 1e-3) / 1e9);
  // the consumer's extra stream: xg (8 B) + 4 B of index per far element, read once more by the SpMV kernel
  const long long extra16 = far * 12 / 16;
  t = time_it(6, e0, e1, d_flush, flush_bytes, [&] { hipLaunchKernelGGL(stream_read, dim3(4096), dim3(256), 0, 0, d_xg, extra16, d_out); });
  printf("consumer side: %.1f MB of extra stream (xg + index) read   : %7.1f us alone (a launch; inside the SpMV it costs its bytes: %.1f us at 6.3 TB/s)\n",
         extra16 * 16e-6, t * 1e3, extra16 * 16.0 / 6.3e12 * 1e6);
  return 0;
}
