// any_order_launch_bench.hip -- round 4: the pipelined sharded step cuts a rank's rows into C chunks, one kernel each, so that chunk c can travel while
// chunk c + 1 computes.  On ONE stream every kernel waits for the previous one to drain (the AQL barrier bit HIP sets on every packet of an in-order
// stream: 13 us per boundary on the headline matrix); on TWO streams the kernels overlap, but the hand-overs between streams cost 27 us per step.
// HIP on AMD has an extension for exactly this: hipExtLaunchKernel(..., stopEvent, flags = hipExtAnyOrderLaunch) launches WITHOUT the barrier bit and
// binds an event to the kernel's own completion signal.  This measures, for a streaming kernel cut into C chunks:
//   mode 0: one launch over everything
//   mode 1: C ordinary launches on one stream
//   mode 2: C launches on one stream, chunks 1 .. C-1 with hipExtAnyOrderLaunch
//   mode 3: C launches alternating over two streams (fork / join by events), what the library does today
//   mode 4: mode 2 + a stop event per chunk and a second stream that waits for each and copies 8 B of the chunk's output (the exchange stand-in)
//   mode 5: mode 2 + a stop event per chunk, nothing waiting for them        mode 6: mode 4 without the copies (waits + join only)
//   mode 7: C ordinary launches on one stream, hipEventRecord behind each, the second stream waiting for each (no copies)
// and checks that (a) the event of chunk c really is behind chunk c's writes (the copy behind it sees them) and (b) an ordinary launch after the
// any-order ones waits for all of them.
// Usage: any_order_launch_bench [MB=650] [C=8]
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double double2v __attribute__((ext_vector_type(2)));
constexpr int kBlockBytes = 8192; // 256 lanes x 2 x 16 B

// block b reads 8 KB and writes one double: out[b] = stamp + (sum == impossible ? 1 : 0)
__global__ __launch_bounds__(256) void rd(const double *__restrict__ a, long long first_block, double stamp, double *__restrict__ out) {
  const long long b = first_block + blockIdx.x;
  const double2v *p = reinterpret_cast<const double2v *>(a) + b * (kBlockBytes / 16);
  const double2v u = __builtin_nontemporal_load(p + threadIdx.x), w = __builtin_nontemporal_load(p + 256 + threadIdx.x);
  double s = u.x + u.y + w.x + w.y;
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[b] = stamp + ((part[0] + part[1] + part[2] + part[3]) == 123.456 ? 1.0 : 0.0);
}
__global__ void after(const double *out, long long nblocks, double stamp, int *bad) { // an ORDINARY launch behind the chunks: must see every out[b]
  for (long long b = blockIdx.x * static_cast<long long>(blockDim.x) + threadIdx.x; b < nblocks; b += static_cast<long long>(gridDim.x) * blockDim.x)
    if (out[b] != stamp) atomicAdd(bad, 1);
}

#define CHECK(x)                                                                                  \
  do {                                                                                            \
    hipError_t e_ = (x);                                                                          \
    if (e_ != hipSuccess) {                                                                       \
      std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                                  \
      return 1;                                                                                   \
    }                                                                                             \
  } while (0)

int main(int argc, char **argv) {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const double mb = argc > 1 ? atof(argv[1]) : 650.0;
  const int C = argc > 2 ? atoi(argv[2]) : 8;
  const long long nblocks = static_cast<long long>(mb * 1e6 / kBlockBytes) / C * C, per = nblocks / C;
  double *a = nullptr, *out = nullptr, *probe = nullptr;
  int *bad = nullptr;
  CHECK(hipMalloc(reinterpret_cast<void **>(&a), nblocks * kBlockBytes));
  CHECK(hipMalloc(reinterpret_cast<void **>(&out), nblocks * sizeof(double)));
  CHECK(hipMalloc(reinterpret_cast<void **>(&probe), C * sizeof(double)));
  CHECK(hipMalloc(reinterpret_cast<void **>(&bad), sizeof(int)));
  CHECK(hipMemset(a, 0, nblocks * kBlockBytes));
  hipStream_t s0, s1, sx;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&sx, hipStreamNonBlocking));
  hipEvent_t t0, t1, fork, join1, joinx;
  CHECK(hipEventCreate(&t0));
  CHECK(hipEventCreate(&t1));
  CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CHECK(hipEventCreateWithFlags(&join1, hipEventDisableTiming));
  CHECK(hipEventCreateWithFlags(&joinx, hipEventDisableTiming));
  std::vector<hipEvent_t> done(C);
  for (auto &e : done) CHECK(hipEventCreate(&e)); // (stop events of hipExtLaunchKernel carry timestamps)
  std::printf("%lld blocks of 8 KB (%.0f MB), C = %d chunks of %lld blocks\n", nblocks, nblocks * 8192e-6, C, per);

  double stamp = 0.0;
  auto launch = [&](hipStream_t st, long long first, long long count, bool any_order, hipEvent_t stop) -> hipError_t {
    if (!any_order && !stop) {
      hipLaunchKernelGGL(rd, dim3(static_cast<unsigned>(count)), dim3(256), 0, st, a, first, stamp, out);
      return hipGetLastError();
    }
    void *args[] = {&a, &first, &stamp, &out};
    return hipExtLaunchKernel(reinterpret_cast<const void *>(rd), dim3(static_cast<unsigned>(count)), dim3(256), args, 0, st, nullptr, stop,
                              any_order ? hipExtAnyOrderLaunch : 0);
  };
  auto run_mode = [&](int mode) -> hipError_t {
    hipError_t e = hipSuccess;
    auto keep = [&](hipError_t r) { if (e == hipSuccess) e = r; };
    if (mode == 0) keep(launch(s0, 0, nblocks, false, nullptr));
    if (mode == 1)
      for (int c = 0; c < C; ++c) keep(launch(s0, c * per, per, false, nullptr));
    if (mode == 2)
      for (int c = 0; c < C; ++c) keep(launch(s0, c * per, per, c > 0, nullptr));
    if (mode == 3) {
      keep(hipEventRecord(fork, s0));
      keep(hipStreamWaitEvent(s1, fork, 0));
      for (int c = 0; c < C; ++c) keep(launch(c & 1 ? s1 : s0, c * per, per, false, nullptr));
      keep(hipEventRecord(join1, s1));
      keep(hipStreamWaitEvent(s0, join1, 0));
    }
    if (mode == 4 || mode == 6) {
      for (int c = 0; c < C; ++c) {
        keep(launch(s0, c * per, per, c > 0, done[c]));
        keep(hipStreamWaitEvent(sx, done[c], 0));
        if (mode == 4) keep(hipMemcpyAsync(probe + c, out + (c + 1) * per - 1, sizeof(double), hipMemcpyDeviceToDevice, sx)); // the chunk's LAST block's output
      }
      keep(hipEventRecord(joinx, sx));
      keep(hipStreamWaitEvent(s0, joinx, 0));
    }
    if (mode == 5)
      for (int c = 0; c < C; ++c) keep(launch(s0, c * per, per, c > 0, done[c]));
    if (mode == 7) {
      for (int c = 0; c < C; ++c) {
        keep(launch(s0, c * per, per, false, nullptr));
        keep(hipEventRecord(done[c], s0));
        keep(hipStreamWaitEvent(sx, done[c], 0));
      }
      keep(hipEventRecord(joinx, sx));
      keep(hipStreamWaitEvent(s0, joinx, 0));
    }
    return e;
  };
  const char *names[] = {"one launch", "C launches, one stream", "C launches, one stream, any-order", "C launches over two streams (fork / join)",
                         "any-order + stop events + exchange stream", "any-order + stop events, nobody waits", "any-order + stop events + waits, no copies",
                         "ordinary launches + hipEventRecord + waits"};
  for (int mode = 0; mode < 8; ++mode) {
    float best = 1e30f, sum = 0.f;
    int wrong_after = 0, wrong_probe = 0;
    const int reps = 30;
    for (int r = -5; r < reps; ++r) {
      stamp += 1.0;
      CHECK(hipMemsetAsync(bad, 0, sizeof(int), s0));
      CHECK(hipMemsetAsync(probe, 0, C * sizeof(double), s0));
      CHECK(hipStreamSynchronize(s0));
      CHECK(hipEventRecord(t0, s0));
      const hipError_t e = run_mode(mode);
      if (e != hipSuccess) {
        std::printf("mode %d: %s\n", mode, hipGetErrorString(e));
        return 1;
      }
      CHECK(hipEventRecord(t1, s0));
      hipLaunchKernelGGL(after, dim3(1024), dim3(256), 0, s0, out, nblocks, stamp, bad);
      CHECK(hipStreamSynchronize(s0));
      CHECK(hipStreamSynchronize(sx));
      int h_bad = 0;
      CHECK(hipMemcpy(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost));
      wrong_after += h_bad;
      if (mode == 4) {
        std::vector<double> h(C);
        CHECK(hipMemcpy(h.data(), probe, C * sizeof(double), hipMemcpyDeviceToHost));
        for (int c = 0; c < C; ++c) wrong_probe += h[c] != stamp;
      }
      float ms = 0.f;
      CHECK(hipEventElapsedTime(&ms, t0, t1));
      if (r >= 0) {
        best = ms < best ? ms : best;
        sum += ms;
      }
    }
    std::printf("mode %d  %-46s: mean %7.1f us  min %7.1f us   stale outputs seen by the ordinary launch behind: %d%s\n", mode, names[mode],
                sum / reps * 1e3, best * 1e3, wrong_after, mode == 4 ? (wrong_probe ? "   EXCHANGE COPIES RAN EARLY" : "   exchange copies saw their chunk") : "");
  }
  return 0;
}
