// k_hub.hip -- hub rows through L2-sized column windows (round 4; tunable hub_rows).  The second level of the column-slab blocking of
// k_segment.hip, for matrices with power-law rows AND columns (R-MAT): the few rows that hold most of the non-zeros are cut, without a copy,
// into runs over column windows small enough that one window's part of x (4 MB by default) fits an XCD's L2, and one launch walks all
// those runs window by window; the many short rows, whose runs would be one or two non-zeros long at that granularity, stay with the
// coarse slab passes (k_segment.hip, which then skips the hub rows).
//
// Reference role: none.  The reference's answer to a power-law matrix is row-block splitting (hip-csr-adaptive-plus/csr_adaptive_plus_analyze.cpp:45-63:
// long rows get dedicated blocks) and every block gathers from all of x; on MI355X the gather side is what bounds such a matrix
// (tools/micro/xcd_slab_gather_bench.hip: 512 M gathers with R-MAT columns take 7.6 ms anywhere in a 256 MB x, 4.9 ms with eight-way slabs,
// 2.5-2.7 ms when every gather stays inside a 2-4 MB window -- profiles/r04_xcd_window_gather_microbench.txt).
// Why rows are split by LENGTH: a row of d non-zeros spread over W windows yields min(d, ~W) runs; runs cost 12 B of list, a partial sum and the
// part-used cache lines at both ends, so windows pay only where d >> W.  On R-MAT scale 25 the rows of >= 128 non-zeros are 3 % of the rows and
// 80 % of the non-zeros (analytic model in DESIGN.md section 3).
//
// Plan (structure only -- no values, no columns; the stale-plan guard covers the row structure like every other table):
//   hub_row[h]            the hub rows, ascending
//   per entry e           ent_hub (index into hub_row), ent_begin (first non-zero), vptr (prefix of lengths): the runs in (window, hub row) order,
//                         runs longer than kSegPiece cut into pieces, exactly the entry format of k_segment.hip -- the SAME tile kernel stages them
//   blk_first[b]          first entry of workgroup b (cut by cost like the coarse passes)
//   cell[w][g]            first entry of window w whose hub index is >= g * kHubGroup: lets the merge walk a group of hub rows window by window
// SpMV: segment_tile_kernel over all entries writes one partial sum per entry (ys[e] = alpha * sum, streaming, no y access, no atomics);
// hub_merge_kernel adds a hub row's partial sums in window order -- a workgroup per kHubGroup hub rows, its accumulators in LDS, one barrier per
// window -- and adds the result to y[hub_row[h]] once.  Deterministic: the order of every addition is fixed by the plan.
#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

// flag[r] = row r is a hub (>= min_len non-zeros); flag[m] = 0 closes the scan
__global__ __launch_bounds__(kThreads) void hub_flag_kernel(const int *__restrict__ rp, int m, int min_len, int *__restrict__ flag) {
  const long long r = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (r < m) flag[r] = (rp[r + 1] - rp[r] >= min_len) ? 1 : 0;
  else if (r == m) flag[r] = 0;
}
__global__ __launch_bounds__(kThreads) void hub_rows_kernel(const int *__restrict__ flag, const int *__restrict__ pos, int m, int *__restrict__ hub_row) {
  const long long r = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (r < m && flag[r]) hub_row[pos[r]] = static_cast<int>(r);
}
// every hub row's columns must not decrease (the windows of a row are then contiguous runs): one wavefront per hub row, strided
__global__ __launch_bounds__(kThreads) void hub_sorted_kernel(const int *__restrict__ rp, const int *__restrict__ ci, const int *__restrict__ hub_row, int H,
                                                              int *__restrict__ unsorted) {
  const int lane = threadIdx.x & (kWave - 1);
  for (long long h = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave; h < H;
       h += static_cast<long long>(gridDim.x) * (kThreads / kWave)) {
    const int r = hub_row[h];
    const int j0 = rp[r], j1 = rp[r + 1];
    bool bad = false;
    for (int j = j0 + lane; j + 1 < j1; j += kWave) bad = bad || ci[j] > ci[j + 1];
    if (__ballot(bad) != 0ULL && lane == 0) atomicOr(unsorted, 1);
  }
}
// wbeg[w * H + h], w = 0 .. NW: first non-zero of hub row h whose column is >= w * width (w = NW: the row's end)
__global__ __launch_bounds__(kThreads) void hub_bounds_kernel(const int *__restrict__ rp, const int *__restrict__ ci, const int *__restrict__ hub_row, int H,
                                                              int width, int NW, int *__restrict__ wbeg) {
  const long long t = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (t >= static_cast<long long>(NW + 1) * H) return;
  const int w = static_cast<int>(t / H), h = static_cast<int>(t - static_cast<long long>(w) * H);
  const int r = hub_row[h];
  int lo = rp[r], hi = rp[r + 1];
  if (w == 0) hi = lo;
  else if (w == NW) lo = hi;
  else {
    const long long key = static_cast<long long>(w) * width; // first column of window w
    while (lo < hi) {
      const int mid = lo + (hi - lo) / 2;
      if (ci[mid] >= key) hi = mid; else lo = mid + 1;
    }
  }
  wbeg[t] = lo;
}
// pieces[w * H + h] = entries the run (w, h) contributes (0 for an empty run); pieces[NW * H] = 0 closes the scan
__global__ __launch_bounds__(kThreads) void hub_pieces_kernel(const int *__restrict__ wbeg, int H, int NW, int piece_max, int *__restrict__ pieces) {
  const long long t = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  const long long cells = static_cast<long long>(NW) * H;
  if (t < cells) pieces[t] = (wbeg[t + H] - wbeg[t] + piece_max - 1) / piece_max;
  else if (t == cells) pieces[t] = 0;
}
__global__ __launch_bounds__(kThreads) void hub_compact_kernel(const int *__restrict__ wbeg, const int *__restrict__ pos, int H, int NW, int piece_max,
                                                               int *__restrict__ ent_hub, int *__restrict__ ent_begin, int *__restrict__ ent_len) {
  const long long t = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (t >= static_cast<long long>(NW) * H) return;
  int at = wbeg[t];
  int left = wbeg[t + H] - at;
  const int h = static_cast<int>(t % H);
  for (int i = pos[t]; left > 0; ++i) { // (the longest hub row of R-MAT 25 holds ~100 K non-zeros in one window: ~200 entries)
    const int len = left < piece_max ? left : piece_max;
    ent_hub[i] = h;
    ent_begin[i] = at;
    ent_len[i] = len;
    at += len;
    left -= len;
  }
}
// cell[w * (NG + 1) + g] = first entry of window w with hub index >= g * kHubGroup (g = NG: the window's end)
__global__ __launch_bounds__(kThreads) void hub_cells_kernel(const int *__restrict__ pos, int H, int NW, int NG, int *__restrict__ cell) {
  const long long t = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (t >= static_cast<long long>(NW) * (NG + 1)) return;
  const int w = static_cast<int>(t / (NG + 1)), g = static_cast<int>(t - static_cast<long long>(w) * (NG + 1));
  const long long h = static_cast<long long>(g) * kHubGroup < H ? static_cast<long long>(g) * kHubGroup : H;
  cell[t] = pos[static_cast<long long>(w) * H + h]; // (w * H + H is window w + 1's first cell: pos has NW * H + 1 entries)
}

// y[hub_row[h]] += the row's partial sums, added in window order (pieces of a cut run in entry order): a workgroup per kHubGroup hub rows
__global__ __launch_bounds__(kThreads) void hub_merge_kernel(int H, int NW, int NG, const int *__restrict__ cell, const int *__restrict__ ent_hub,
                                                             const double *__restrict__ ys, const int *__restrict__ hub_row, double *__restrict__ y) {
  static_assert(kHubGroup == kThreads, "one accumulator per lane");
  const int g = blockIdx.x, t = threadIdx.x;
  const int h0 = g * kHubGroup;
  __shared__ double acc[kHubGroup];
  acc[t] = 0.0;
  __syncthreads();
  const int *c = cell + g;
  int e0 = c[0], e1 = c[1];
  for (int w = 0; w < NW; ++w) { // (workgroup-uniform)
    // next window's cell while this one's entries are in flight
    int n0 = 0, n1 = 0;
    if (w + 1 < NW) {
      n0 = c[static_cast<size_t>(w + 1) * (NG + 1)];
      n1 = c[static_cast<size_t>(w + 1) * (NG + 1) + 1];
    }
    for (int e = e0 + t; e < e1; e += kThreads) {
      const int h = ent_hub[e];
      if (e == e0 || ent_hub[e - 1] != h) { // the first piece of the run (w, h) adds all of them: one writer per accumulator per window
        double s = ys[e];
        for (int k = e + 1; k < e1 && ent_hub[k] == h; ++k) s += ys[k];
        acc[h - h0] += s;
      }
    }
    __syncthreads(); // the next window's writers may be other lanes
    e0 = n0;
    e1 = n1;
  }
  const int h = h0 + t;
  if (h < H) {
    const int r = hub_row[h];
    y[r] += acc[t];
  }
}

} // namespace

void launch_hub_flags(hipStream_t stream, const int *rp, int m, int min_len, int *flag) {
  const long long blocks = (static_cast<long long>(m) + 1 + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(hub_flag_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, rp, m, min_len, flag);
}
void launch_hub_rows(hipStream_t stream, const int *flag, const int *pos, int m, int *hub_row) {
  if (m <= 0) return;
  const long long blocks = (static_cast<long long>(m) + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(hub_rows_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, flag, pos, m, hub_row);
}
void launch_hub_sorted(hipStream_t stream, const CsrDev &A, const int *hub_row, int H, int *unsorted) {
  if (H <= 0) return;
  long long blocks = (static_cast<long long>(H) + (kThreads / kWave) - 1) / (kThreads / kWave);
  if (blocks > 65521) blocks = 65521; // (prime: see kMaxGridBlocks)
  hipLaunchKernelGGL(hub_sorted_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, A.rp, A.ci, hub_row, H, unsorted);
}
void launch_hub_bounds(hipStream_t stream, const CsrDev &A, const int *hub_row, int H, int width, int NW, int *wbeg) {
  const long long total = static_cast<long long>(NW + 1) * H;
  if (total <= 0) return;
  hipLaunchKernelGGL(hub_bounds_kernel, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, A.rp, A.ci, hub_row, H,
                     width, NW, wbeg);
}
void launch_hub_pieces(hipStream_t stream, const int *wbeg, int H, int NW, int piece_max, int *pieces) {
  const long long total = static_cast<long long>(NW) * H + 1;
  hipLaunchKernelGGL(hub_pieces_kernel, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, wbeg, H, NW, piece_max,
                     pieces);
}
void launch_hub_compact(hipStream_t stream, const int *wbeg, const int *pos, int H, int NW, int piece_max, int *ent_hub, int *ent_begin, int *ent_len) {
  const long long total = static_cast<long long>(NW) * H;
  if (total <= 0) return;
  hipLaunchKernelGGL(hub_compact_kernel, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, wbeg, pos, H, NW,
                     piece_max, ent_hub, ent_begin, ent_len);
}
void launch_hub_cells(hipStream_t stream, const int *pos, int H, int NW, int NG, int *cell) {
  const long long total = static_cast<long long>(NW) * (NG + 1);
  if (total <= 0) return;
  hipLaunchKernelGGL(hub_cells_kernel, dim3(static_cast<unsigned>((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, pos, H, NW, NG, cell);
}
void launch_hub_merge(hipStream_t stream, int H, int NW, int NG, const int *cell, const int *ent_hub, const double *ys, const int *hub_row, double *y) {
  if (H <= 0 || NG <= 0) return;
  hipLaunchKernelGGL(hub_merge_kernel, dim3(NG), dim3(kThreads), 0, stream, H, NW, NG, cell, ent_hub, ys, hub_row, y);
}

} // namespace spmv_acc
