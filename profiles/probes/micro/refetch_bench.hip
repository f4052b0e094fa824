// refetch_bench.hip -- round 4: the slab passes over run lists fetch every 128-B line of the medium rows once per pass that owns a run in it (~5x on
// R-MAT 25), each time from HBM, because the passes are seconds... milliseconds apart.  Would the SAME re-fetches be cheaper if the passes were ordered
// band-major -- all slabs of a 30 MB band of the matrix before the next band, the slabs on different XCDs -- so that the re-fetches hit the 256 MB
// Infinity Cache instead of HBM?  This reads a 6 GB array R times:
//   mode A: R sequential launches, each reading everything once (what the passes do today): every read comes from HBM
//   mode B: ONE launch, band-major: the R reads of a band follow one another within microseconds and come from R different XCDs
//           (block g: band = g / (R * BB), rep = (g % (R * BB)) / BB, chunk = band * BB + (g + rep) % BB  -> the reps of a chunk sit on different XCDs)
//   mode C: mode B with all reps of a chunk on the SAME XCD (chunk = band * BB + g % BB, BB a multiple of 8): re-fetches hit that XCD's L2 if it still holds the line
// Usage: refetch_bench [GB=6] [R=5] [band_MB=32]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double double2v __attribute__((ext_vector_type(2)));
constexpr int kBlockBytes = 8192; // 256 lanes x 2 x 16 B

template <int MODE> __global__ __launch_bounds__(256) void rd(const double *__restrict__ a, long long nblocks, int R, long long BB, double *out) {
  long long chunk;
  if (MODE == 0) {
    chunk = blockIdx.x;
  } else {
    const long long g = blockIdx.x;
    const long long band = g / (R * BB), in = g % (R * BB), rep = in / BB;
    chunk = band * BB + (MODE == 1 ? (in + rep) % BB : in % BB);
  }
  if (chunk >= nblocks) return;
  const double2v *p = reinterpret_cast<const double2v *>(a) + chunk * (kBlockBytes / 16);
  const double2v u = __builtin_nontemporal_load(p + threadIdx.x), w = __builtin_nontemporal_load(p + 256 + threadIdx.x);
  const double s = u.x + u.y + w.x + w.y;
  if (s == 123.456) out[0] = s;
}

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const double gb = argc > 1 ? atof(argv[1]) : 6.0;
  const int R = argc > 2 ? atoi(argv[2]) : 5;
  const int band_mb = argc > 3 ? atoi(argv[3]) : 32;
  const long long nblocks = static_cast<long long>(gb * 1e9 / kBlockBytes) / 8 * 8;
  long long BB = static_cast<long long>(band_mb) * 1048576 / kBlockBytes / 8 * 8;
  const long long bands = (nblocks + BB - 1) / BB;
  double *a, *out;
  if (hipMalloc(&a, nblocks * kBlockBytes) != hipSuccess || hipMalloc(&out, 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0, nblocks * kBlockBytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const double total = static_cast<double>(nblocks) * kBlockBytes * R;
  printf("%.2f GB array, read %d times = %.1f GB; bands of %d MB (%lld blocks of 8 KB), %lld bands\n", nblocks * kBlockBytes / 1e9, R, total / 1e9, band_mb, BB, bands);
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) {
        for (int r = 0; r < R; ++r) hipLaunchKernelGGL(rd<0>, dim3(static_cast<unsigned>(nblocks)), dim3(256), 0, 0, a, nblocks, R, BB, out);
      } else {
        const long long grid = bands * R * BB; // (< 2^24 workgroups for the defaults: 6 GB / 8 KB x 5 = 3.7 M)
        if (mode == 1) hipLaunchKernelGGL(rd<1>, dim3(static_cast<unsigned>(grid)), dim3(256), 0, 0, a, nblocks, R, BB, out);
        else hipLaunchKernelGGL(rd<2>, dim3(static_cast<unsigned>(grid)), dim3(256), 0, 0, a, nblocks, R, BB, out);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep > 0 && ms < best) best = ms;
    }
    const char *names[] = {"A: R sequential full passes (every read from HBM)", "B: one launch, band-major, the reps of a chunk on different XCDs (re-fetches from the Infinity Cache)",
                           "C: one launch, band-major, the reps of a chunk on the same XCD (re-fetches from that L2 while it holds them)"};
    printf("mode %-110s: %8.1f us  %6.2f TB/s of reads\n", names[mode], best * 1e3, total / (best * 1e-3) / 1e12);
  }
  return 0;
}
