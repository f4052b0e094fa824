// k_segment.hip -- column-slab blocking WITHOUT a copy of the matrix (tunable slab_segments): where every row's columns ascend,
// the non-zeros of row r that fall into column slab s are ONE contiguous run of the caller's arrays.  The plan keeps, per slab, the
// list of those runs ("segments": row, first non-zero, length -- 12 B each, structure only), and an SpMV is S passes, pass s summing
// every segment of slab s straight from the caller's colindex / values and gathering from 1/S of x.
//
// Reference role: none (the reference streams every row against all of x; SURVEY.md section 7 names the problem -- random 8-B reads
// of a 256 MB x on R-MAT -- and "column-window blocking" as the thing to consider).  k_slab.hip is the first form of the same idea:
// it re-orders colindex and values into slab-major COPIES (plan = 12 B per non-zero, values refreshed by hand after an in-place
// edit), which is why it is opt-in.  Here the plan holds no values and no columns: a caller who edits values in place is served
// the new values, and an in-place edit of COLUMNS can only move gathers to another slab than the one they were filed under --
// the sums are taken over the same non-zeros whatever the slab boundaries were, so the result stays right and only the locality
// degrades.  The row structure is covered by the plan guard like every other table.
//
// Price: a pass reads runs of ~12 non-zeros separated by the other slabs' runs, so the streams use a part of every 128-B line
// they touch (short rows of a power-law matrix lie mostly inside one slab, which is what keeps that affordable), and every
// run is one more read-modify-write of a y element (the copy form pays the same for its compact slabs).
//
// Build: a row's non-zeros are counted per slab -- rows of more than 32 by one wavefront each (ballots, as k_slab.hip), shorter ones by one LANE
// each -- and the slab index is checked never to decrease along the row; the exclusive prefix of the counts is where each slab's run starts.  Runs longer than
// kSegPiece are cut into pieces (entries of their own); pieces -> scan -> compact as in k_slab.hip; the entries' lengths are
// scanned into `vptr` (an entry's place in the pass's VIRTUAL non-zero order: the runs laid end to end), and the pass is cut into
// workgroups by cost (max(length, 4) per entry, kSegCost per workgroup: at most 2047 products and 511 entries each).
// SpMV pass (segment_tile_kernel): a workgroup owns consecutive entries = a contiguous piece of the virtual order, which it
// stages as ONE 2048-product LDS tile exactly like the other tile kernels -- lane t takes virtual non-zeros t, t + 256, ..., so
// a wavefront's loads are as coalesced as the runs are long -- after finding every element's entry: each entry marks its first
// element in a 2-byte owner map, a max-scan over the map fills the rest, address = run start + offset in the run.  Then w lanes
// per entry sum its span of the tile (tile_row_sum, long spans to whole waves); a whole run adds alpha * sum straight into y, a piece
// of a cut run writes it to ys[e].  (The first form, w lanes
// walking each run straight from global memory, ran R-MAT 25 in 6.8-10 ms against 7.1 for the default path: runs are power-law
// long, every wavefront waits for its longest, and cutting them shorter only added entries.)  Merge (slabs that hold cut runs only): the first
// piece of a run adds the run's partial sums to y in entry order (the pieces are consecutive entries: deterministic, no atomics).
#include "device_utils.hpp"
#include "kernels.hpp"
#include "tile_stage.hpp"

namespace spmv_acc {
namespace {

using namespace dev;

constexpr int kSegShortRow = 32;                    // rows up to this long are counted one lane per row at plan time
constexpr int kSegTile = kThreads * kNnzPerThread; // 2048 products
constexpr int kSegCost = 1536;                      // cost per workgroup; + one entry of at most kSegPiece = at most 2047 products
constexpr int kSegMinCost = 4;                      // an entry costs at least this much: at most 511 entries per workgroup
constexpr int kSegEntries = 512;
static_assert(kSegCost + kSegPiece <= kSegTile && (kSegCost + kSegPiece) / kSegMinCost <= kSegEntries, "a workgroup's entries fit one tile");
static_assert(kNnzPerThread == 8, "the owner map is scanned 8 elements per lane (one 16-B LDS word)");

// cnt[s][row] (S x (m + 1), entry m zeroed for the scans), beg[s][row] (same shape): first non-zero of slab s's run in row `row`
__global__ __launch_bounds__(kThreads) void segment_count_kernel(const int *__restrict__ rp, const int *__restrict__ ci, int m, SlabBounds B,
                                                                 int S, int *__restrict__ cnt, int *__restrict__ beg,
                                                                 int *__restrict__ not_monotone, int rest_below) {
  const int lane = threadIdx.x & (kWave - 1);
  const int SC = rest_below > 0 ? S - 1 : S; // column slabs; with rest_below the last plane (S - 1) lists the rows of < rest_below non-zeros WHOLE
  // One wavefront per row; beyond kMaxGridBlocks * 4 rows the wavefronts stride over the rows -- by a PRIME number of workgroups: a stride of 2^16
  // wavefronts gave wavefront 0 the rows k * 2^16, on R-MAT (a row's length falls with the number of set bits in its index) all hubs, 6.6 M
  // non-zeros walked by one wavefront: 96 ms for the kernel, where the longest row alone takes 9
  for (long long row = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave; row < m;
       row += static_cast<long long>(gridDim.x) * (kThreads / kWave)) {
    const int j0 = rp[row], j1 = rp[row + 1];
    if (j1 - j0 <= kSegShortRow) continue; // (wave-uniform: short rows are counted 64 to a wavefront by segment_count_short_kernel)
    if (j1 - j0 < rest_below) { // a short row of the two-class form: one run, all columns, in the last plane (its columns need no order)
      if (lane < S) {
        cnt[static_cast<size_t>(lane) * (static_cast<size_t>(m) + 1) + row] = lane == S - 1 ? j1 - j0 : 0;
        beg[static_cast<size_t>(lane) * (static_cast<size_t>(m) + 1) + row] = j0;
      }
      continue;
    }
    int acc = 0;
    int prev_last = 0; // slab of the last non-zero of the previous step
    bool bad = false;
    for (int base = j0; base < j1; base += kWave) {
      const int j = base + lane;
      const bool live = j < j1;
      int slab = S; // (dead lanes: above every slab, so the order test passes)
      if (live) {
        const int c = ci[j];
        slab = 0;
#pragma unroll
        for (int b = 0; b < 15; ++b) slab += (b < SC - 1 && c >= B.first[b]) ? 1 : 0; // first[b] = first column of slab b + 1
      }
      const int left = __shfl_up(slab, 1, kWave);
      bad = bad || (live && slab < (lane == 0 ? prev_last : left));
      const int last_live = (j1 - base < kWave ? j1 - base : kWave) - 1;
      prev_last = __shfl(slab, last_live, kWave);
      for (int s = 0; s < S; ++s) {
        const unsigned long long mask = __ballot(slab == s);
        if (lane == s) acc += __popcll(mask);
      }
    }
    if (__ballot(bad) != 0ULL && lane == 0) atomicOr(not_monotone, 1);
    // exclusive prefix of the counts over lanes 0 .. S-1 (S <= 16)
    int incl = lane < S ? acc : 0;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int up = __shfl_up(incl, off, kWave);
      if (lane >= off) incl += up;
    }
    if (lane < S) {
      cnt[static_cast<size_t>(lane) * (static_cast<size_t>(m) + 1) + row] = acc;
      beg[static_cast<size_t>(lane) * (static_cast<size_t>(m) + 1) + row] = j0 + incl - acc;
    }
  }
}

// The same for rows of at most kSegShortRow non-zeros, one LANE per row (power-law matrices: 33.5 M rows of R-MAT 25 average 16 non-zeros, and one
// wavefront per row spent 89 ms on them; neighbouring lanes read neighbouring rows, i.e. the same cache lines).  Counts are packed 8 bits per
// slab into two 64-bit words (a row of <= 32 non-zeros cannot overflow a field).
__global__ __launch_bounds__(kThreads) void segment_count_short_kernel(const int *__restrict__ rp, const int *__restrict__ ci, int m, SlabBounds B,
                                                                       int S, int *__restrict__ cnt, int *__restrict__ beg,
                                                                       int *__restrict__ not_monotone, int rest_below) {
  const int SC = rest_below > 0 ? S - 1 : S;
  const long long row = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (row == m) { // (the grid covers m + 1 entries: the scans' closing zeros)
    for (int s = 0; s < S; ++s) cnt[static_cast<size_t>(s) * (static_cast<size_t>(m) + 1) + m] = 0;
  }
  if (row >= m) return;
  const int j0 = rp[row], j1 = rp[row + 1];
  if (j1 - j0 > kSegShortRow) return; // a long row: segment_count_kernel
  if (j1 - j0 < rest_below) { // two-class form: the whole row is one run of the last plane
    for (int s = 0; s < S; ++s) {
      cnt[static_cast<size_t>(s) * (static_cast<size_t>(m) + 1) + row] = s == S - 1 ? j1 - j0 : 0;
      beg[static_cast<size_t>(s) * (static_cast<size_t>(m) + 1) + row] = j0;
    }
    return;
  }
  unsigned long long p0 = 0ULL, p1 = 0ULL; // slabs 0-7, 8-15
  int prev = 0;
  bool bad = false;
  for (int j = j0; j < j1; ++j) {
    const int c = ci[j];
    int slab = 0;
#pragma unroll
    for (int b = 0; b < 15; ++b) slab += (b < SC - 1 && c >= B.first[b]) ? 1 : 0;
    bad = bad || slab < prev;
    prev = slab;
    const unsigned long long inc = 1ULL << (8 * (slab & 7));
    if (slab < 8) p0 += inc;
    else p1 += inc;
  }
  if (bad) atomicOr(not_monotone, 1);
  int at = j0;
  for (int s = 0; s < S; ++s) {
    const int n = static_cast<int>(((s < 8 ? p0 : p1) >> (8 * (s & 7))) & 0xFFULL);
    cnt[static_cast<size_t>(s) * (static_cast<size_t>(m) + 1) + row] = n;
    beg[static_cast<size_t>(s) * (static_cast<size_t>(m) + 1) + row] = at;
    at += n;
  }
}

// pieces[r] = entries row r contributes to this slab's list (0, or its run cut into pieces of at most piece_max); pieces[m] = 0
__global__ __launch_bounds__(kThreads) void segment_pieces_kernel(const int *__restrict__ cnt_s, int m, int piece_max, int *__restrict__ pieces) {
  const long long r = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (r < m) pieces[r] = (cnt_s[r] + piece_max - 1) / piece_max;
  else if (r == m) pieces[r] = 0;
}

__global__ __launch_bounds__(kThreads) void segment_compact_kernel(const int *__restrict__ cnt_s, const int *__restrict__ beg_s,
                                                                   const int *__restrict__ pos, int m, int piece_max,
                                                                   int *__restrict__ seg_row, int *__restrict__ seg_begin,
                                                                   int *__restrict__ seg_len, int *__restrict__ has_pieces) {
  const long long r = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (r >= m) return;
  int left = cnt_s[r];
  int at = beg_s[r];
  // a run cut into pieces: its entries carry the row with the top bit set (their sums go through ys and the merge kernel)
  const int tag = left > piece_max ? static_cast<int>(static_cast<unsigned>(r) | 0x80000000u) : static_cast<int>(r);
  if (left > piece_max) atomicAdd(has_pieces, 1); // (a count: the merge list is sized from it)
  for (int i = pos[r]; left > 0; ++i) { // (a hub row writes a few hundred entries)
    const int len = left < piece_max ? left : piece_max;
    seg_row[i] = tag;
    seg_begin[i] = at;
    seg_len[i] = len;
    at += len;
    left -= len;
  }
}

// the pass's cut runs, for the merge kernel: cut[k] = first entry of a run that was cut into pieces.  In whatever order the atomic hands out the
// places: every cut run is merged by itself, so the order of the list does not reach the sums
__global__ __launch_bounds__(kThreads) void segment_cut_list_kernel(const int *__restrict__ cnt_s, const int *__restrict__ pos, int m, int piece_max,
                                                                    int *__restrict__ counter, int *__restrict__ cut) {
  const long long r = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (r < m && cnt_s[r] > piece_max) cut[atomicAdd(counter, 1)] = pos[r];
}

// cost[i] = max(len[i], kSegMinCost) for the workgroup cut; both arrays get a closing zero for the scans
__global__ __launch_bounds__(kThreads) void segment_cost_kernel(int entries, int *__restrict__ seg_len, int *__restrict__ cost) {
  const long long i = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (i < entries) cost[i] = seg_len[i] > kSegMinCost ? seg_len[i] : kSegMinCost;
  else if (i == entries) cost[i] = seg_len[i] = 0;
}
// blk_first[b] = first entry whose cost prefix is >= b * kSegCost (b = 0 .. nblocks - 1), blk_first[nblocks] = entries
__global__ __launch_bounds__(kThreads) void segment_blocks_kernel(int entries, int nblocks, const int *__restrict__ cptr, int *__restrict__ blk_first) {
  const long long b = static_cast<long long>(blockIdx.x) * kThreads + threadIdx.x;
  if (b > nblocks) return;
  if (b == nblocks) {
    blk_first[b] = entries;
    return;
  }
  const long long want = b * kSegCost;
  int lo = 0, hi = entries; // first e in [0, entries] with cptr[e] >= want (cptr ascends strictly)
  while (lo < hi) {
    const int mid = lo + (hi - lo) / 2;
    if (cptr[mid] >= want) hi = mid; else lo = mid + 1;
  }
  blk_first[b] = lo;
}

// one pass: y[row] += alpha * sum over the run (ys[e] = alpha * sum for the pieces of a long run), a workgroup per kSegCost of cost
template <bool HINT>
__global__ __launch_bounds__(kThreads) void segment_tile_kernel(double alpha, const int *__restrict__ blk_first, const int *__restrict__ seg_row,
                                                                const int *__restrict__ seg_begin, const int *__restrict__ vptr,
                                                                const int *__restrict__ ci, const double *__restrict__ v,
                                                                const double *__restrict__ x, double *__restrict__ ys, double *__restrict__ y,
                                                                const unsigned char *__restrict__ cold) {
  __shared__ __attribute__((aligned(16))) double tile[kSegTile];
  // the owner map lives in the tile's first 4 KB: it is read into registers (and a barrier passed) before the first product is
  // written.  21.9 KB of LDS per workgroup = 7 workgroups per CU, like the other tile kernels (28 KB with a map of its own: 5)
  unsigned short *owner = reinterpret_cast<unsigned short *>(tile);
  __shared__ int s_begin[kSegEntries];
  __shared__ int s_ofs[kSegEntries + 1];
  __shared__ int wave_top[kThreads / kWave];
  __shared__ TileSpans spans;
  const int t = threadIdx.x;
  const int e0 = blk_first[blockIdx.x], e1 = blk_first[blockIdx.x + 1];
  const int ne = e1 - e0;          // 1 .. 511
  const int base = vptr[e0];
  const int total = vptr[e1] - base; // 1 .. 2047
  typedef unsigned int uint4v __attribute__((ext_vector_type(4)));
  reinterpret_cast<uint4v *>(owner)[t] = uint4v{0u, 0u, 0u, 0u};
  for (int i = t; i <= ne; i += kThreads) {
    s_ofs[i] = vptr[e0 + i] - base;
    if (i < ne) s_begin[i] = seg_begin[e0 + i];
  }
  if (t == 0) spans.n = 0; // published by the barriers below, long before tile_row_sum looks at it
  __syncthreads();
  for (int i = t; i < ne; i += kThreads) owner[s_ofs[i]] = static_cast<unsigned short>(i); // every entry holds >= 1 element: distinct places
  __syncthreads();
  // inclusive max-scan over the owner map: lane t takes elements 8 t .. 8 t + 7
  {
    const uint4v q = reinterpret_cast<const uint4v *>(owner)[t];
    unsigned o[8] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16, q.z & 0xFFFFu, q.z >> 16, q.w & 0xFFFFu, q.w >> 16};
#pragma unroll
    for (int j = 1; j < 8; ++j) o[j] = o[j] > o[j - 1] ? o[j] : o[j - 1];
    int incl = static_cast<int>(o[7]);
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
      const int up = __shfl_up(incl, off, kWave);
      if ((t & (kWave - 1)) >= off) incl = incl > up ? incl : up;
    }
    if ((t & (kWave - 1)) == kWave - 1) wave_top[t / kWave] = incl;
    int before = __shfl_up(incl, 1, kWave);
    if ((t & (kWave - 1)) == 0) before = 0;
    __syncthreads();
    for (int wv = 0; wv < t / kWave; ++wv) before = before > wave_top[wv] ? before : wave_top[wv];
    const unsigned pre = static_cast<unsigned>(before);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = o[j] > pre ? o[j] : pre;
    reinterpret_cast<uint4v *>(owner)[t] = uint4v{o[0] | (o[1] << 16), o[2] | (o[3] << 16), o[4] | (o[5] << 16), o[6] | (o[7] << 16)};
  }
  __syncthreads();
  // stage: virtual non-zero k = i * 256 + t lives at run start + offset within the run; all stream loads, then all gathers
  int cc[kNnzPerThread];
  double aa[kNnzPerThread];
  unsigned cb = 0; // HINT: bit i = the plan's cold bit of this lane's i-th non-zero
#pragma unroll
  for (int i = 0; i < kNnzPerThread; ++i) {
    const int k = i * kThreads + t;
    cc[i] = -1;
    aa[i] = 0.0;
    if (k < total) {
      const int e = owner[k];
      const int at = s_begin[e] + (k - s_ofs[e]);
      cc[i] = load_stream(ci + at);
      aa[i] = load_stream(v + at);
      if (HINT) cb |= ((static_cast<unsigned>(cold[at >> 3]) >> (at & 7)) & 1u) << i;
    }
  }
  __syncthreads(); // every lane has read its owners: the map's place becomes tile
  const XGather xr = make_xgather(x, HINT);
#pragma unroll
  for (int i = 0; i < kNnzPerThread; ++i)
    if (cc[i] >= 0) tile[i * kThreads + t] = aa[i] * (HINT ? gather_hinted(xr, cc[i], (cb >> i) & 1u) : x[cc[i]]);
  __syncthreads();
  // sums: w lanes per entry, up to 256 entries per round
  for (int r0 = 0; r0 < ne; r0 += kThreads) { // (workgroup-uniform)
    const int nr = ne - r0 < kThreads ? ne - r0 : kThreads;
    int w = 64;
    while (w > 1 && nr * w > kThreads) w >>= 1;
    const int lane = t & (w - 1);
    const int id = t / w;
    const bool live = id < nr;
    const int lo = live ? s_ofs[r0 + id] : 0;
    const int hi = live ? s_ofs[r0 + id + 1] : 0;
    double acc = tile_row_sum<kThreads>(tile, spans, lo, hi, lane, w);
    acc = group_sum_dyn(acc, w);
    if (live && lane == 0) {
      // a whole run is the only entry of its row in this pass: straight into y (beta was applied before the first pass); a piece
      // of a long run goes through ys, the merge kernel adds a row's pieces in order
      const int r = seg_row[e0 + r0 + id];
      if (r >= 0) y[r] += alpha * acc;
      else ys[e0 + r0 + id] = alpha * acc;
    }
  }
}

// the pieces of long runs: y[row] += their partial sums.  One WAVEFRONT per cut run (the plan's list of them): lane l adds the pieces l, l + 64, ...
// in that order, a fixed butterfly adds the lanes -- the same order every run: deterministic, no atomics.  (Until round 4 one THREAD per entry looked
// for first pieces and added its run's pieces one by one: 18 us per pass on R-MAT 25 just to read 5 M tags, and 30-90 ms on a matrix whose single
// giant row is 586 K pieces -- profiles/r04_extreme_shapes_check.txt.)
__global__ __launch_bounds__(kThreads) void segment_merge_kernel(int ncut, const int *__restrict__ cut, int entries, const int *__restrict__ seg_row,
                                                                 const double *__restrict__ ys, double *__restrict__ y) {
  const int lane = threadIdx.x & (kWave - 1);
  const long long k = static_cast<long long>(blockIdx.x) * (kThreads / kWave) + threadIdx.x / kWave;
  if (k >= ncut) return; // (wave-uniform)
  const int first = cut[k];
  const int r = seg_row[first];
  double sum = 0.0;
  for (long long base = first;; base += kWave) { // the run's pieces are consecutive entries with the same tagged row
    const long long j = base + lane;
    const bool mine = j < entries && seg_row[j] == r;
    if (mine) sum += ys[j];
    if (__ballot(mine) != ~0ULL) break;
  }
  sum = group_sum<64>(sum);
  if (lane == 0) y[static_cast<int>(static_cast<unsigned>(r) & 0x7FFFFFFFu)] += sum;
}

} // namespace

bool launch_segment_count(hipStream_t stream, const CsrDev &A, const SlabBounds &B, int S, int *cnt, int *beg, int *not_monotone, int rest_below) {
  if (S < 1 || S > kSegMaxPlanes) return false; // (one counter per plane in 16 lanes / 16 packed bytes)
  if (A.m <= 0) return true;
  long long blocks = (static_cast<long long>(A.m) + (kThreads / kWave) - 1) / (kThreads / kWave); // one wavefront per row ...
  if (blocks > max_grid_blocks()) blocks = max_grid_blocks();                                         // ... up to what a launch holds
  SPMV_ACC_LAUNCH(segment_count_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, A.rp, A.ci, A.m, B, S, cnt,
                     beg, not_monotone, rest_below);
  const long long short_blocks = (static_cast<long long>(A.m) + 1 + kThreads - 1) / kThreads; // m + 1: the closing zeros
  SPMV_ACC_LAUNCH(segment_count_short_kernel, dim3(static_cast<unsigned>(short_blocks)), dim3(kThreads), 0, stream, A.rp, A.ci, A.m, B, S,
                     cnt, beg, not_monotone, rest_below);
  return true;
}
void launch_segment_pieces(hipStream_t stream, const int *cnt_s, int m, int piece_max, int *pieces) {
  const long long blocks = (static_cast<long long>(m) + 1 + kThreads - 1) / kThreads;
  SPMV_ACC_LAUNCH(segment_pieces_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, cnt_s, m, piece_max, pieces);
}
void launch_segment_compact(hipStream_t stream, const int *cnt_s, const int *beg_s, const int *pos, int m, int piece_max, int *seg_row,
                            int *seg_begin, int *seg_len, int *has_pieces) {
  if (m <= 0) return;
  const long long blocks = (static_cast<long long>(m) + kThreads - 1) / kThreads;
  SPMV_ACC_LAUNCH(segment_compact_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, cnt_s, beg_s, pos, m, piece_max,
                     seg_row, seg_begin, seg_len, has_pieces);
}
void launch_segment_cost(hipStream_t stream, int entries, int *seg_len, int *cost) {
  const long long blocks = (static_cast<long long>(entries) + 1 + kThreads - 1) / kThreads;
  SPMV_ACC_LAUNCH(segment_cost_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, entries, seg_len, cost);
}
int segment_block_count(long long total_cost) { return static_cast<int>((total_cost + kSegCost - 1) / kSegCost); }
void launch_segment_blocks(hipStream_t stream, int entries, int nblocks, const int *cptr, int *blk_first) {
  const long long blocks = (static_cast<long long>(nblocks) + 1 + kThreads - 1) / kThreads;
  SPMV_ACC_LAUNCH(segment_blocks_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, entries, nblocks, cptr, blk_first);
}
void launch_segment_tiles(hipStream_t stream, int nblocks, double alpha, const int *blk_first, const int *seg_row, const int *seg_begin,
                          const int *vptr, const int *ci, const double *v, const double *x, double *ys, double *y, const unsigned char *cold) {
  if (nblocks <= 0) return;
  if (cold)
    SPMV_ACC_LAUNCH(segment_tile_kernel<true>, dim3(nblocks), dim3(kThreads), 0, stream, alpha, blk_first, seg_row, seg_begin, vptr, ci, v, x, ys, y, cold);
  else
    SPMV_ACC_LAUNCH(segment_tile_kernel<false>, dim3(nblocks), dim3(kThreads), 0, stream, alpha, blk_first, seg_row, seg_begin, vptr, ci, v, x, ys, y, cold);
}
void launch_segment_cut_list(hipStream_t stream, const int *cnt_s, const int *pos, int m, int piece_max, int *counter, int *cut) {
  if (m <= 0) return;
  const long long blocks = (static_cast<long long>(m) + kThreads - 1) / kThreads;
  SPMV_ACC_LAUNCH(segment_cut_list_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, cnt_s, pos, m, piece_max, counter, cut);
}
void launch_segment_merge(hipStream_t stream, int ncut, const int *cut, int entries, const int *seg_row, const double *ys, double *y) {
  if (ncut <= 0) return;
  const long long blocks = (static_cast<long long>(ncut) + (kThreads / kWave) - 1) / (kThreads / kWave);
  SPMV_ACC_LAUNCH(segment_merge_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, stream, ncut, cut, entries, seg_row, ys, y);
}

} // namespace spmv_acc
