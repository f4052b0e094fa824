// k_slab.hip -- OPT-IN column-slab blocking (tunable col_slabs): A = sum_s A_s, A_s = the non-zeros whose column lies in the
// s-th of S equal column ranges, y = beta*y + alpha * sum_s A_s x as S consecutive SpMVs of the ordinary kernels.
//
// Reference role: none -- the reference streams every row against all of x.  SURVEY.md section 7 names the problem ("R-MAT ... x =
// 256 MB, random 8 B reads per 64/128 B line ... consider column-window blocking"): on power-law columns ~60 % of the gathers fall
// on a few MB of x, but every one of the eight 4 MB L2s has to hold that hot set AND absorb the lines the cold gathers bring in.
// With S slabs a phase gathers from 1/S of x only: the L2s hold a hot set S times deeper (profiles/probes/micro/xcd_slab_gather_bench.hip:
// 67 -> 109 G gathers/s with an eighth of x per L2; profiles/probes/col_slab_probe.py: R-MAT scale 25 7.15 -> 5.46 ms with S = 8 DENSE slabs,
// i.e. S passes over y and S full row-pointer arrays).  The slabs are kept COMPACT -- only the rows that have non-zeros in a slab,
// with a row-id list (profiles/probes/col_slab_compact_probe.py: 45 M non-empty (row, slab) pairs of 268 M on R-MAT 25) -- so a phase is an
// ordinary SpMV of a smaller matrix into a compact y_s plus a merge y[rowid] += y_s.  The plan then holds a re-ordered COPY of colindex and values, which is why this is
// opt-in like the 16-bit column stream: after editing VALUES in place call spmv_acc_refresh_values (one scatter pass, the slabs'
// structure and plans stay), after editing the structure spmv_acc_release_plans.
//
// Build (once per plan, order inside a row preserved, no atomics): one wavefront per row, 64 non-zeros per step;
//   count  : per step one ballot per slab, lane s keeps slab s's count            -> cnt[s][row]
//   scan   : exclusive scan of each slab's counts (rocPRIM, tuner.cpp)           -> rowptr_s
//   scatter: the same walk; a non-zero's place = rowptr_s[row] + non-zeros of slab s before it in the row (ballot + popcount).
#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

__global__ __launch_bounds__(kThreads) void slab_count_kernel(const int *__restrict__ rp, const int *__restrict__ ci, int m, int width,
                                                              int S, int *__restrict__ cnt) {
  const int lane = threadIdx.x & (kWave - 1);
  // (one wavefront per row, no grid stride: a stride of 2^16 wavefronts hands wavefront 0 the rows k * 2^16, which on R-MAT are all hubs --
  // 82 ms for this kernel and 106 for the scatter where the longest row alone takes 9; see k_segment.hip)
  for (long long row = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave; row < m;
       row += static_cast<long long>(gridDim.x) * (kThreads / kWave)) { // (the stride: kMaxGridBlocks, a prime, and only beyond 33.5 M rows)
    const int j0 = rp[row], j1 = rp[row + 1];
    int acc = 0;
    for (int base = j0; base < j1; base += kWave) {
      const int j = base + lane;
      const int slab = j < j1 ? ci[j] / width : -1;
      for (int s = 0; s < S; ++s) {
        const unsigned long long mask = __ballot(slab == s);
        if (lane == s) acc += __popcll(mask);
      }
    }
    if (lane < S) cnt[static_cast<size_t>(lane) * (static_cast<size_t>(m) + 1) + row] = acc;
  }
  // the scan runs over m + 1 entries per slab: the last one is the total
  if (blockIdx.x == 0 && static_cast<int>(threadIdx.x) < S) cnt[static_cast<size_t>(threadIdx.x) * (static_cast<size_t>(m) + 1) + m] = 0;
}

// off[s] = first position of slab s in the re-ordered arrays (exclusive sums of the slabs' non-zero counts)
__global__ __launch_bounds__(kThreads) void slab_scatter_kernel(const int *__restrict__ rp, const int *__restrict__ ci,
                                                                const double *__restrict__ v, int m, int width, int S,
                                                                const int *__restrict__ rps, const long long *__restrict__ off,
                                                                int *__restrict__ ci_out, double *__restrict__ v_out, int values_only) {
  const int lane = threadIdx.x & (kWave - 1);
  for (long long row = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave; row < m;
       row += static_cast<long long>(gridDim.x) * (kThreads / kWave)) { // one wavefront per row, striding beyond kMaxGridBlocks workgroups
    const int j0 = rp[row], j1 = rp[row + 1];
    long long pos = 0; // lane s: where slab s's next non-zero of this row goes
    if (lane < S) pos = off[lane] + rps[static_cast<size_t>(lane) * (static_cast<size_t>(m) + 1) + row];
    for (int base = j0; base < j1; base += kWave) {
      const int j = base + lane;
      const bool live = j < j1;
      const int c = live ? ci[j] : 0;
      const double a = live ? v[j] : 0.0;
      const int slab = live ? c / width : -1;
      for (int s = 0; s < S; ++s) {
        const unsigned long long mask = __ballot(slab == s);
        const long long at = __shfl(pos, s, kWave);
        if (slab == s) {
          const long long dst = at + __popcll(mask & ((1ULL << lane) - 1ULL));
          if (!values_only) ci_out[dst] = c; // (values_only: the caller changed values in place and asked for a refresh)
          v_out[dst] = a;
        }
        if (lane == s) pos += __popcll(mask);
      }
    }
  }
}

// A slab keeps only the rows that HAVE non-zeros in it (power-law matrices: R-MAT 25 in 8 slabs has 45 M non-empty (row, slab)
// pairs out of 268 M): flags -> exclusive scan (tuner.cpp) -> row-id list + the row pointers of those rows.
// rps: the slab's DENSE row pointers (m + 1); flags / pos: m + 1 ints.
__global__ __launch_bounds__(kThreads) void slab_flags_kernel(const int *__restrict__ rps, int m, int *__restrict__ flags) {
  const long long r = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (r < m) flags[r] = rps[r + 1] > rps[r] ? 1 : 0;
  else if (r == m) flags[r] = 0;
}
__global__ __launch_bounds__(kThreads) void slab_compact_kernel(const int *__restrict__ rps, const int *__restrict__ pos, int m,
                                                                int *__restrict__ rowid, int *__restrict__ crp) {
  const long long r = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (r < m) {
    if (rps[r + 1] > rps[r]) {
      const int i = pos[r];
      rowid[i] = static_cast<int>(r);
      crp[i] = rps[r];
    }
  } else if (r == m) {
    crp[pos[m]] = rps[m]; // the slab's non-zero count closes the compact row pointers
  }
}
// y[rowid[i]] += ys[i]: a slab's compact result into the full vector (rows ascending: the stores are as coalesced as the rows are dense)
__global__ __launch_bounds__(kThreads) void slab_merge_kernel(int ms, const int *__restrict__ rowid, const double *__restrict__ ys,
                                                              double *__restrict__ y) {
  const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < ms) {
    const int r = rowid[i];
    y[r] += ys[i];
  }
}

} // namespace

void launch_slab_flags(hipStream_t stream, const int *rps, int m, int *flags) {
  const long long blocks = (static_cast<long long>(m) + 1 + kThreads - 1) / kThreads;
  SPMV_ACC_LAUNCH(slab_flags_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, rps, m, flags);
}
void launch_slab_compact(hipStream_t stream, const int *rps, const int *pos, int m, int *rowid, int *crp) {
  const long long blocks = (static_cast<long long>(m) + 1 + kThreads - 1) / kThreads;
  SPMV_ACC_LAUNCH(slab_compact_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, rps, pos, m, rowid, crp);
}
void launch_slab_merge(hipStream_t stream, int ms, const int *rowid, const double *ys, double *y) {
  if (ms <= 0) return;
  const long long blocks = (static_cast<long long>(ms) + kThreads - 1) / kThreads;
  SPMV_ACC_LAUNCH(slab_merge_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, ms, rowid, ys, y);
}

void launch_slab_count(hipStream_t stream, const CsrDev &A, int width, int S, int *cnt) {
  if (A.m <= 0) return;
  long long blocks = (static_cast<long long>(A.m) + (kThreads / kWave) - 1) / (kThreads / kWave); // one wavefront per row ...
  if (blocks > max_grid_blocks()) blocks = max_grid_blocks();                                         // ... up to what a launch holds
  SPMV_ACC_LAUNCH(slab_count_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, A.rp, A.ci, A.m, width, S, cnt);
}

void launch_slab_scatter(hipStream_t stream, const CsrDev &A, int width, int S, const int *rps, const long long *off, int *ci_out,
                         double *v_out, bool values_only) {
  if (A.m <= 0) return;
  long long blocks = (static_cast<long long>(A.m) + (kThreads / kWave) - 1) / (kThreads / kWave);
  if (blocks > max_grid_blocks()) blocks = max_grid_blocks();
  SPMV_ACC_LAUNCH(slab_scatter_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, A.rp, A.ci, A.v, A.m, width, S,
                     rps, off, ci_out, v_out, values_only ? 1 : 0);
}


// ---- value samples of a plan that holds a copy of the values (the automatic slab-major copy, dispatch.cpp::slab_copy_auto) -------------------------
// The caller may edit VALUES in place between two SpMVs (every other plan survives that: they stream values from the caller's array).  A plan
// that holds a copy compares `count` evenly spaced samples of the caller's values with what they were when the copy was made, before it uses
// the copy; *changed (pinned host memory) is raised on the first difference (bitwise comparison: NaNs compare by their bits).
namespace {
// saved: count value samples (64-bit patterns) followed by count colindex samples (32 bits each, two to a word).  *changed: bit 0 = a value differs,
// bit 1 = a column index differs (the copy's structure is stale: the engine drops it).
__global__ __launch_bounds__(256) void value_samples_kernel(const double *__restrict__ v, const int *__restrict__ ci, long long lo, long long span, int count,
                                                            unsigned long long *__restrict__ saved, int *__restrict__ changed) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= count) return;
  const long long j = lo + (count > 1 ? static_cast<long long>(k) * span / (count - 1) : 0);
  const unsigned long long bits = static_cast<unsigned long long>(__double_as_longlong(v[j]));
  int *saved_ci = reinterpret_cast<int *>(saved + count);
  const int c = ci[j];
  if (changed == nullptr) { // record
    saved[k] = bits;
    saved_ci[k] = c;
  } else {
    if (saved[k] != bits) __hip_atomic_fetch_or(changed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (saved_ci[k] != c) __hip_atomic_fetch_or(changed, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
} // namespace
void launch_value_samples(hipStream_t stream, const double *v, const int *ci, long long lo, long long span, int count, unsigned long long *saved,
                          int *changed) {
  if (count <= 0) return;
  SPMV_ACC_LAUNCH(value_samples_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, v, ci, lo, span, count, saved, changed);
}

} // namespace spmv_acc
