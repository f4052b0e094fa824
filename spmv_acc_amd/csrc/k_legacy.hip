// k_legacy.hip -- genuine kernels for the two KERNEL_STRATEGY names that were aliases until round 3: LIGHT and BLOCK_ROW_ORDINARY.
// ("first as aliases to the nearest kernel, then as genuine ... kernels".  Neither is on BASELINE's path; they are
// part of the plugin surface and of the benchmark's symbol list, benchmark_spmv_acc.hpp:62,47.)
//
// Reference roles:
//   * hip-light/spmv_hip_acc_imp.inl:36-76 + light_spmv.cpp:16-41 (LightSpMV, doi 10.1007/s11265-016-1216-4): a fixed grid of
//     wavefronts; each wavefront fetches its next rows from ONE global counter with an atomic add ("dynamic row distribution"), a
//     vector of w lanes sums one row, w chosen from the average row length.  The reference allocates, zeroes and frees the counter on
//     every call and gives each fetch 64 / w rows.
//   * hip-block-row-ordinary/spmv_hip_acc_imp.cpp:16-66: one 256-thread block per row, a fixed grid of 64 blocks striding over the
//     rows, a 256-entry LDS tree per row with commented-out barriers between its levels (SURVEY.md A.3 lists it among the latent races).
// What is kept: the work distribution that defines each strategy -- rows handed out by an atomic counter; a whole workgroup per row.
// What is different (MI355X-first): the counter lives in the plan and the kernel leaves it at zero itself -- the last wavefront
// to finish resets it (a second counter tells which one is last) -- so a call is ONE launch with no malloc / memset / free around it
// and replays from a hipGraph (a memset node in front of the kernel was the first form: under torch's graph capture the replayed
// kernel found the counter unreset and did nothing); a fetch takes a batch of 8 * (64 / w) rows so that a lane group keeps 8 rows' loads in flight
// and the counter sees 1/8 of the atomics; the grid is sized from the device (CUs x resident waves), not a constant 256 x 256;
// general alpha / beta and the out-of-place old y like every other kernel.  Block-per-row: 16-B stream loads, 4 non-zeros per lane per
// step, DPP wave sums and ONE barrier per row (4 partials through LDS) instead of an 8-level tree.
// Both are the simple kernels their names promise: no imbalance rescue, no tiles.  On the matrices they suit (LIGHT: very uneven row
// lengths at moderate size; block-per-row: rows of thousands of non-zeros) they are within reach of the tile kernels; elsewhere they are
// what the reference's are -- baselines.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

constexpr int kLightRowsPerGroup = 8; // rows a lane group takes from one fetch (their loads are issued together)

// w lanes per row (wave-uniform, power of two <= 64).  counter[0]: rows handed out so far; counter[1]: wavefronts that have finished.
// Both are zero when a launch starts and when it ends.
__global__ __launch_bounds__(kThreads) void light_kernel(int m, int w, double alpha, double beta, unsigned *__restrict__ counter,
                                                         const int *__restrict__ rp, const int *__restrict__ ci,
                                                         const double *__restrict__ v, const double *__restrict__ x, double *y,
                                                         const double *yin, const int *__restrict__ guard, int *__restrict__ stale) {
  check_plan_guard(rp, m, guard, stale);
  const int lane_in_wave = threadIdx.x & (kWave - 1);
  const int lane = lane_in_wave & (w - 1);
  const int group = lane_in_wave / w;
  const int groups = kWave / w;
  const int batch = groups * kLightRowsPerGroup; // rows per fetch
  for (;;) {
    // (unsigned: the counter runs past m by up to one batch per resident wavefront, and m may be close to 2^31)
    unsigned fetched = 0;
    if (lane_in_wave == 0) fetched = atomicAdd(counter, static_cast<unsigned>(batch));
    fetched = __shfl(fetched, 0, kWave);
    if (fetched >= static_cast<unsigned>(m)) break; // every wavefront reaches this: counter[0] only grows while any wavefront runs
    const int base = static_cast<int>(fetched);
    // rows base + group + k * groups, k = 0 .. 7: neighbouring groups read neighbouring rowptr entries
    int j0[kLightRowsPerGroup], j1[kLightRowsPerGroup];
#pragma unroll
    for (int k = 0; k < kLightRowsPerGroup; ++k) {
      const long long row = static_cast<long long>(base) + group + k * groups;
      j0[k] = j1[k] = 0;
      if (row < m) {
        j0[k] = rp[row];
        j1[k] = rp[row + 1];
      }
    }
    // (written out: a loop over k around loops of different trip counts is not unrolled, and s[] would then live in scratch)
    auto row_sum = [&](int a, int b) {
      double acc = 0.0;
      for (int j = a + lane; j < b; j += w) acc += load_stream(v + j) * x[load_stream(ci + j)];
      return acc;
    };
    static_assert(kLightRowsPerGroup == 8, "eight rows per lane group are written out below");
    const double s[kLightRowsPerGroup] = {row_sum(j0[0], j1[0]), row_sum(j0[1], j1[1]), row_sum(j0[2], j1[2]), row_sum(j0[3], j1[3]),
                                          row_sum(j0[4], j1[4]), row_sum(j0[5], j1[5]), row_sum(j0[6], j1[6]), row_sum(j0[7], j1[7])};
    for (int k = 0; k < kLightRowsPerGroup; ++k) {
      const double sum = group_sum_dyn(s[k], w); // every lane takes part (DPP needs a full exec mask)
      const long long row = static_cast<long long>(base) + group + k * groups;
      if (row < m && lane == 0) store_y(y, yin, static_cast<int>(row), alpha, beta, sum);
    }
  }
  // the last wavefront out puts both counters back to zero for the next launch (every other wavefront has made its last fetch)
  if (lane_in_wave == 0) {
    const unsigned waves = gridDim.x * (kThreads / kWave);
    if (atomicAdd(counter + 1, 1u) == waves - 1) {
      counter[0] = 0;
      counter[1] = 0;
    }
  }
}

// One workgroup per row, grid-striding over the rows.  4 consecutive non-zeros per lane per step (16-B loads), the row start aligned
// down to a multiple of 4 and the foreign elements at both ends masked out of the sum.
__global__ __launch_bounds__(kThreads) void block_row_kernel(int m, int nnz, double alpha, double beta, const int *__restrict__ rp,
                                                             const int *__restrict__ ci, const double *__restrict__ v,
                                                             const double *__restrict__ x, double *y, const double *yin,
                                                             const int *__restrict__ guard, int *__restrict__ stale) {
  check_plan_guard(rp, m, guard, stale);
  constexpr int kWaves = kThreads / kWave;
  __shared__ double part[2][kWaves]; // double-buffered: the next row's partials do not wait for this row's reader
  int flip = 0;
  for (long long row_ll = blockIdx.x; row_ll < m; row_ll += gridDim.x, flip ^= 1) {
    const int row = static_cast<int>(row_ll);
    const int j0 = rp[row], j1 = rp[row + 1];
    double s = 0.0;
    for (int base = (j0 & ~3) + 4 * static_cast<int>(threadIdx.x); base < j1; base += 4 * kThreads) {
      if (base + 4 <= nnz) {
        const int4v c = load_stream_i4<true>(ci + base);
        const double2v a0 = load_stream_d2<true>(v + base);
        const double2v a1 = load_stream_d2<true>(v + base + 2);
        const double p0 = a0.x * x[c.x], p1 = a0.y * x[c.y], p2 = a1.x * x[c.z], p3 = a1.y * x[c.w];
        s += (base + 0 >= j0 && base + 0 < j1) ? p0 : 0.0;
        s += (base + 1 >= j0 && base + 1 < j1) ? p1 : 0.0;
        s += (base + 2 >= j0 && base + 2 < j1) ? p2 : 0.0;
        s += (base + 3 >= j0 && base + 3 < j1) ? p3 : 0.0;
      } else {
        for (int e = 0; e < 4; ++e)
          if (base + e >= j0 && base + e < j1) s += v[base + e] * x[ci[base + e]];
      }
    }
    s = group_sum<64>(s);
    if ((threadIdx.x & (kWave - 1)) == 0) part[flip][threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
      double total = 0.0;
#pragma unroll
      for (int i = 0; i < kWaves; ++i) total += part[flip][i];
      store_y(y, yin, row, alpha, beta, total);
    }
  }
}

} // namespace

void launch_light(hipStream_t stream, const CsrDev &A, int w, int grid_blocks, unsigned *counter, double alpha, double beta, const double *x,
                  double *y) {
  if (A.m <= 0) return;
  SPMV_ACC_LAUNCH(light_kernel, dim3(grid_blocks), dim3(kThreads), 0, stream, A.m, w, alpha, beta, counter, A.rp, A.ci, A.v, x, y,
                     A.yin ? A.yin : y, A.guard, A.stale);
}

void launch_block_row(hipStream_t stream, const CsrDev &A, int grid_blocks, double alpha, double beta, const double *x, double *y) {
  if (A.m <= 0) return;
  // (an ODD number of workgroups: with a power-of-two grid workgroup 0 would walk the rows k * grid, which on matrices whose row length follows
  // the bits of the row index -- R-MAT -- are all hubs)
  const int grid = A.m < grid_blocks ? A.m : (grid_blocks | 1);
  SPMV_ACC_LAUNCH(block_row_kernel, dim3(grid), dim3(kThreads), 0, stream, A.m, A.nnz, alpha, beta, A.rp, A.ci, A.v, x, y,
                     A.yin ? A.yin : y, A.guard, A.stale);
}

} // namespace spmv_acc
