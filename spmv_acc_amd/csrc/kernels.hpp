// kernels.hpp -- host-side launchers of the gfx950 kernels (definitions in k_*.hip).
// Launchers only enqueue work on `stream`; they never allocate, copy or synchronise, so a caller
// may capture them into a hipGraph.
#pragma once

#include <hip/hip_runtime_api.h>
#include <hip/hip_ext.h>

namespace spmv_acc {

// ---- kernel clock (round 5) --------------------------------------------------------------------------------------------------------------
// Every launcher launches through SPMV_ACC_LAUNCH.  Ordinarily that IS hipLaunchKernelGGL.  While the calling host thread's kernel clock is on
// (spmv_acc_time_spmv_kernels, c_api.cpp) each launch carries its own start / stop event pair -- hipExtLaunchKernelGGL writes the dispatch's own
// begin / end timestamps into them, the figures rocprofv3 --kernel-trace reports -- so that a call's KERNEL time (the sum over its launches) can be
// read live, beside the event-pair time of the reference harness's protocol, which also holds the protocol's floor (marker packets, dispatch latency).
bool kernel_clock_next(hipEvent_t *start, hipEvent_t *stop); // false: the clock is off (the usual case: one thread-local load)
#define SPMV_ACC_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                                  \
  do {                                                                                                                           \
    hipEvent_t kc_e0_ = nullptr, kc_e1_ = nullptr;                                                                               \
    if (::spmv_acc::kernel_clock_next(&kc_e0_, &kc_e1_))                                                                         \
      hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, kc_e0_, kc_e1_, 0, __VA_ARGS__);                                 \
    else                                                                                                                         \
      hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                                       \
  } while (0)

// Device-side CSR view (all pointers are device pointers; int32 indices, fp64 values).
struct CsrDev {
  int m = 0;
  int n = 0;
  int nnz = 0;  // END offset of the view's non-zeros in ci / v: rowptr[m] (the non-zero COUNT only where rowptr[0] == 0)
  int nnz0 = 0; // rowptr[0]: 0 for a whole matrix, the first non-zero of an un-rebased row sub-range (shard.cpp's chunk views, `rowptr + r0`)
  int count() const { return nnz - nnz0; } // what every shape heuristic, tile range, census and budget reads (round 5; they read nnz until then)
  const int *rp = nullptr;
  const int *ci = nullptr;
  const double *v = nullptr;
  bool aligned16 = false; // ci and v are 16-byte aligned and nnz >= 8: wide-load kernels allowed
  // Stale-plan guard (plan.cpp): kGuardSamples rowptr entries recorded when the plan was built (device array) and a
  // sticky flag in pinned host memory.  The first wave of block 0 of every SpMV kernel re-reads the samples and raises
  // the flag when the matrix behind these pointers is no longer the one the plan was built for.  Null: no check.
  const int *guard = nullptr;
  int *stale = nullptr;
  // Gather hints (tuner.cpp ensure_hint, k_hint.hip): one bit per non-zero, set where the non-zero's x line is NOT among the
  // hot lines that fit an L2; the tile kernels (row blocks, row-block-plus, flat) issue those gathers non-temporal.  Null: no hints (set per launch by the
  // engine: only while the plan's timed comparison says they pay, and only where x is below 4 GB).  Speed only.
  const unsigned char *cold = nullptr;
  // Out-of-place call (spmv_acc_csr_spmv_oop): where the old y is read; null = the y that is written (every reference entry).
  // Set per call by the engine, under the plan's lock; every launcher passes `yin ? yin : y` to its kernels.
  const double *yin = nullptr;
};
// 32-bit gather offsets (device_utils.hpp gather_u32): legal while every column's byte offset 8 * col fits 32 bits; a view whose column count
// is unknown (n <= 0) keeps the 64-bit form
inline bool x32_ok(const CsrDev &A) { return A.n > 0 && A.n <= (1 << 29); }
constexpr int kGuardSamples = 64; // rowptr[k * m / 63], k = 0 .. 63 (includes rowptr[0] and rowptr[m] = nnz)
void launch_guard_fill(hipStream_t stream, const int *rp, int m, int *d_guard);
void launch_guard_check(hipStream_t stream, const CsrDev &A); // the check alone, for paths whose SpMV kernels run on derived matrices
// column-slab blocking without a copy (k_segment.hip, tunable slab_segments): cnt S x (m + 1), beg S x (m + 1); *not_monotone pre-zeroed
struct SlabBounds {
  int first[15]; // first[b] = first column of slab b + 1 (ascending); slab of column c = number of entries <= c among the first S - 1
};
constexpr int kSegMaxPlanes = 16; // planes a build may count (column slabs + the whole-row plane): launch_segment_count refuses more
// rest_below > 0 (two-class form): S counts one plane more than there are column slabs; rows of fewer than rest_below non-zeros are ONE run each, all
// columns, in that last plane (they are not cut by column at all), the longer rows are cut into the S - 1 column slabs as usual
bool launch_segment_count(hipStream_t stream, const CsrDev &A, const SlabBounds &B, int S, int *cnt, int *beg, int *not_monotone, int rest_below = 0);
void launch_segment_pieces(hipStream_t stream, const int *cnt_s, int m, int piece_max, int *pieces);
void launch_segment_compact(hipStream_t stream, const int *cnt_s, const int *beg_s, const int *pos, int m, int piece_max, int *seg_row,
                            int *seg_begin, int *seg_len, int *has_pieces); // *has_pieces pre-zeroed: counts the runs that were cut
// cut[0 .. *counter): first entries of the pass's cut runs (any order); *counter pre-zeroed, cut sized from launch_segment_compact's count
void launch_segment_cut_list(hipStream_t stream, const int *cnt_s, const int *pos, int m, int piece_max, int *counter, int *cut);
constexpr int kSegPiece = 512; // longer runs are cut into pieces (entries of their own)
void launch_segment_cost(hipStream_t stream, int entries, int *seg_len, int *cost); // both entries + 1 long; closes them with a zero
int segment_block_count(long long total_cost);
void launch_segment_blocks(hipStream_t stream, int entries, int nblocks, const int *cptr, int *blk_first); // blk_first: nblocks + 1
void launch_segment_tiles(hipStream_t stream, int nblocks, double alpha, const int *blk_first, const int *seg_row, const int *seg_begin,
                          const int *vptr, const int *ci, const double *v, const double *x, double *ys, double *y, const unsigned char *cold = nullptr);
void launch_segment_merge(hipStream_t stream, int ncut, const int *cut, int entries, const int *seg_row, const double *ys, double *y);
// opt-in full check (k_guard.hip, tunable guard_full): one partial digest of rowptr[0 .. m] per workgroup into part[0 .. parts);
// then ONE workgroup adds them up and either writes the digest to digest_out (plan build) or compares it with `expected` and raises `stale`
constexpr int kDigestMaxParts = 1024;
int rowptr_digest_parts(int m);
void launch_rowptr_digest(hipStream_t stream, const int *rp, int m, unsigned long long *part);
void launch_rowptr_verdict(hipStream_t stream, const unsigned long long *part, int m, unsigned long long expected, int *stale,
                           unsigned long long *digest_out);

// Cache policy of the 16-B colindex / value stream loads of the tile kernels.  Which one is fastest depends on the
// matrix (A/B on MI355X: default-policy loads win by 7-27 % on FEM-like matrices -- part of the matrix then stays
// in the 256 MB Infinity Cache between SpMVs --, non-temporal loads win by 6-13 % where short rows make the x / rowptr
// / y lines in L2 worth protecting), so the engine times the candidates once per matrix and keeps the winner.
constexpr int kStreamPolicyNt = 0;          // both non-temporal
constexpr int kStreamPolicyDefault = 1;     // both default
constexpr int kStreamPolicyIndexDefault = 2; // colindex default, values non-temporal
constexpr int kStreamPolicyValueDefault = 3; // colindex non-temporal, values default

// ---- tile geometry (fixed at build time) -----------------------------------------------------------
constexpr int kThreads = 256;             // 4 waves per workgroup
// HIP launches hold fewer than 2^32 work-items: a grid of 2^24 or more 256-thread workgroups WRAPS (measured on this stack: 70 M rows at one
// wavefront per row = 17.5 M workgroups ran the first 2.9 M rows and reported nothing).  Kernels whose grid grows with m alone stride over a capped,
// prime number of workgroups (prime: a power-of-two stride gives one wavefront all the hub rows of an R-MAT matrix, k_segment.hip).
constexpr int kMaxGridBlocks = 8388593;
int max_grid_blocks(); // = kMaxGridBlocks unless a test lowered it (tunable max_grid_blocks, config.cpp): what the launchers clamp their grids to
constexpr int kNnzPerThread = 8;          // two 4-wide steps per lane per round
constexpr int kTile = kThreads * kNnzPerThread; // 2048 products = 16 KB of LDS per workgroup
constexpr int kVectorTarget = 1900;       // vector-row tile kernel: products a workgroup's rows should bring to its tile (a tunable until round 5: 1900 against
                                          // the row-block family's 1500 is 3-5.5 % faster on four of five sweep stand-ins, equal on the fifth)
constexpr int kPlusThreads = 256;         // row-block-plus ANALYSIS geometry: the reference's (THREADS 256, R 2,
constexpr int kPlusR = 2;                 // MIN_NNZ 1024) instance (csr_adaptive_plus_spmv.cpp:195-202)
constexpr int kPlusMinNnz = 2 * kPlusR * kPlusThreads; // MIN_NNZ_PER_BLOCK 1024
constexpr int kPlusLongChunk = 2 * kPlusMinNnz;        // NN_EI * MIN_NNZ_PER_BLOCK non-zeros per long-row block

// default / vector-row family: `w` lanes per row straight from global memory.
// Rows [0, row_split) use width w0, rows [row_split, m) use width w1 (row_split = m: one width).
void launch_vector_row(hipStream_t stream, const CsrDev &A, int row_split, int w0, int w1, double alpha, double beta,
                       const double *x, double *y, bool single_row_groups = false);

// The same lane layout on the tile machinery (LDS-staged 16-B stream loads; up to four rows per lane group so the tile fills).
// avg0 / avg1: average row length of rows [0, row_split) / [row_split, m) (sizes the rows per workgroup of each half).
void launch_vector_tile(hipStream_t stream, const CsrDev &A, int row_split, int w0, int w1, double avg0, double avg1,
                        int target_products, int xcd_chunk, int stream_policy, double alpha, double beta, const double *x,
                        double *y, bool reverse = false);

// genuine LIGHT (rows handed out by an atomic counter, w lanes per row) and BLOCK_ROW_ORDINARY (one workgroup per row) -- k_legacy.hip.
// counter: two plan-resident unsigneds, zero between launches (the kernel's last wavefront resets them).  grid_blocks: a resident grid (CUs x 8).
void launch_light(hipStream_t stream, const CsrDev &A, int w, int grid_blocks, unsigned *counter, double alpha, double beta, const double *x,
                  double *y);
void launch_block_row(hipStream_t stream, const CsrDev &A, int grid_blocks, double alpha, double beta, const double *x, double *y);

// wavefront-per-row for long rows: 4 consecutive non-zeros per lane per step (16-B loads), two steps in flight.
void launch_wave_row(hipStream_t stream, const CsrDev &A, double alpha, double beta, const double *x, double *y);

// line-enhance family: THREADS/vec consecutive rows per workgroup, non-zeros streamed through an
// LDS tile in rounds.  vec in {1,2,4,8,16,32,64}.
// rows_per_block <= kThreads / vec (0 = kThreads / vec).  flags: bit 0 XCD-contiguous block order, bit 1 read the
// old y at kernel start instead of at the end, bit 2 XCD-chunked block order with chunk = flags >> 8,
// bit 3 per-lane predicated staging (A/B), bits 4-5 cache policy of the stream loads (0 nt, 1 default, 2 colindex default +
// values nt, 3 colindex nt + values default).
// digest (may be null): plan-resident row lengths + per-block bases for THIS rows_per_block, read instead of rowptr.
struct RowDigest {
  int rpb = 0;                          // rows per block the bases were built for
  unsigned char *lens = nullptr;        // m bytes: min(row length, 255)
  int *base = nullptr;                  // nblocks + 1 ints: rowptr[b * rpb]; bit 31 = a row of the block is longer than 255
};
struct Col16; // (below) the plan's 16-bit column encoding: non-null + built = the kernel streams it instead of colindex
void launch_rowblock_stream(hipStream_t stream, const CsrDev &A, int vec, int rows_per_block, int flags,
                            double alpha, double beta, const double *x, double *y, const RowDigest *digest = nullptr,
                            int cache_ends = 0, const Col16 *col16 = nullptr);
void launch_row_digest(hipStream_t stream, const int *rp, int m, int rows_per_block, unsigned char *lens, int *base);
void pick_rowblock_shape(int m, int nnz, int target_products, int *vec, int *rows_per_block);

// plan-time imbalance probe for the row-block family: *d_out (pre-zeroed) = max non-zeros owned by any
// workgroup of rows_per_block consecutive rows.
void launch_max_block_nnz(hipStream_t stream, const int *rp, int m, int rows_per_block, int avg_block, int *d_out); // d_out[2], pre-zeroed
// a row block is "balanced enough" while it needs at most this many LDS rounds
constexpr int kRowblockMaxRounds = 8;

// row-block preprocessing pass, device form: break points with the reference's exact semantics
// (hip-flat/flat_imp.inl:108-131) computed by one binary search per entry; no memset needed.
void launch_break_points(hipStream_t stream, const int *rp, int m, int nnz, int stride, int *bp, int bp_len);

// ---- 16-bit column encoding (k_col16.hip builds it, tile_stage.hpp stage_products_c16 reads it) -----------------------------------
constexpr int kCol16Chunk = 256; // non-zeros per base column: one wavefront's step of the tile kernels (64 lanes x 4), aligned in the absolute non-zero index
struct Col16 {
  int state = -1;                // -1 not looked at, 0 not worth it on this matrix (too many escapes, x of 4 GB or more), 1 built
  int chunk0 = 0;                // the view's first chunk: A.nnz0 rounded down to a flat tile (2048), in chunks (0 unless the matrix is an un-rebased row sub-range); tables are indexed by chunk - chunk0
  int nchunks = 0;
  int rec_ints = 0;              // R: ints per chunk record (16, 32 or 64): {base, escapes, overflow start, 0, first R - 4 escaped columns}
  unsigned short *d16 = nullptr; // nchunks * 256 offsets from the chunk's base; 0xFFFF = escape
  int *rec = nullptr;            // nchunks * R
  int *ovf = nullptr;            // escapes beyond R - 4 per chunk, chunk by chunk (+ 64 entries of padding)
  int *ci_guard = nullptr;       // 64 samples of colindex taken at build time (stale-plan guard of the kernels that no longer read colindex)
  long long escapes = 0;         // all escapes of the view
  long long overflow = 0;        // of them in the overflow list
  size_t bytes = 0;              // what the encoding holds on the device
};
// what a kernel gets (by value): the tables pre-offset so that the kernel indexes them by ABSOLUTE chunk / non-zero index
struct Col16Dev {
  const unsigned short *d16;
  const int *rec;
  const int *ovf;
  const int *ci_guard;
  int rec_ints;
  int guard_lo, guard_span; // colindex[guard_lo + k * guard_span / 63] == ci_guard[k], k = 0 .. 63
};
inline Col16Dev col16_dev(const Col16 &C, const CsrDev &A) {
  Col16Dev d;
  d.d16 = C.d16 - static_cast<long long>(C.chunk0) * kCol16Chunk;
  d.rec = C.rec - static_cast<long long>(C.chunk0) * C.rec_ints;
  d.ovf = C.ovf;
  d.ci_guard = C.ci_guard;
  d.rec_ints = C.rec_ints;
  d.guard_lo = A.nnz0;
  d.guard_span = A.count() > 0 ? A.count() - 1 : 0;
  return d;
}
// gather hints (k_hint.hip)
constexpr int kHintLineShift = 4;       // an x line = 16 columns = 128 B (the L2 line)
constexpr int kHintBins = 4096;         // census histogram: lines / sampled hits per count value, last bin open-ended
constexpr int kHintSamples = 8 << 20;   // non-zeros sampled by the census (all of them below this)
void launch_hint_census(hipStream_t stream, const int *ci, int nnz, int ncols, int stride, int samples, unsigned *counts);
void launch_hint_hist(hipStream_t stream, const unsigned *counts, int nlines, unsigned *hist_lines, unsigned long long *hist_hits);
void launch_hint_bits(hipStream_t stream, const int *ci, int nnz, int ncols, const unsigned *counts, unsigned threshold, unsigned char *bits);

size_t col16_scan_bytes(int nchunks);
void launch_col16_base(hipStream_t stream, const int *ci, int nnz, int chunk0, int nchunks, int *base, int *esc_count, unsigned long long *stats);
void launch_col16_overflow(hipStream_t stream, int *cnt, int nchunks, int E);
bool launch_col16_scan(hipStream_t stream, int nchunks, const int *esc_count, int *esc_start, void *tmp, size_t tmp_bytes);
void launch_col16_encode(hipStream_t stream, const int *ci, int nnz, int chunk0, int nchunks, const int *base, const int *ovf_start, int R,
                         unsigned short *d16, int *rec, int *ovf);
void launch_col16_guard(hipStream_t stream, const int *ci, int lo, int span, int *out);

// flat family: one workgroup per `stride` non-zeros (stride = kThreads * {4, 8, 16}); complete rows are
// stored directly, the two possible partial rows per tile go to head/tail carries that a small second
// kernel folds into y in tile order.
struct FlatPlan {
  int stride = 0;
  int ntiles = 0;
  int tile0 = 0;            // the plan's first tile in absolute tile numbering: A.nnz0 / stride (0 unless the matrix is an un-rebased row sub-range);
                            // tile t of the plan covers non-zeros [(t + tile0) * stride, ...): every per-tile table is indexed by t, only tile origins add it
  int *bp = nullptr;        // ntiles + 1 break points (reference semantics)
  double *head = nullptr;   // per tile: partial sum of the row that started in an earlier tile
  double *tail = nullptr;   // per tile: partial sum of the row that continues in the next tile
  int *tail_row = nullptr;  // per tile: that row, or -1
  int *tail_end = nullptr;  // per tile: rowptr[row + 1] of that row
  void *digest = nullptr;   // per tile, 16 B: {first row, end row, rowptr[end - 1], rowptr[end]} -- what the tile kernel reads instead
                            // of the bp -> rowptr chain (built once per plan from bp and rowptr)
  int xcd_chunk = 0;        // > 0: XCD-chunked tile order (device_utils.hpp::xcd_chunked_block)
  int stream_policy = 0;    // cache policy of the stream loads (kStreamPolicy*)
  bool can_finish = false;  // no row runs more than kFlatFinish non-zeros past the tile it starts in (plan-time probe)
  bool needs_fixup = true;  // true: every cut row is folded from carries by the fix-up kernel; false (only when
                            // can_finish): tiles finish their cut rows themselves.  Chosen by timing, tuner.cpp.
  int max_tile_rows = 0;    // most rows any one tile (= workgroup) owns (plan-time probe)
  bool early_stream = false; // issue the tile's stream loads before the break point -> rowptr chain (small grids, timed)
  bool reverse = false;     // this launch walks the tiles in reverse order (zigzag, set per launch by the engine)
  int cache_ends = 0;       // tiles at each end of the grid that stay cacheable under the non-temporal policy (set per launch)
  bool segment_sum = false; // rows reduced by the segmented scan over the tile (reference option FLAT_SEGMENT_SUM_REDUCE), 2048-tile only
  const Col16 *col16 = nullptr; // columns from the plan's 16-bit encoding instead of colindex (NPT 8 tiles only; set per launch by the engine)
  bool mode_tuned[2] = {false, false};  // per beta class ([0]: beta == 0): tuned_fixup holds the timed choice
  bool tuned_fixup[2] = {true, true};
};
// A tile finishes its last row itself when the row ends at most this many non-zeros past the tile (one wave, two
// unrolled steps).  If any row of the matrix overhangs further, the plan uses head/tail carries and the fix-up kernel
// for ALL cut rows (paying for both the finishing wave and the fix-up launch measured 3-6% slower on the RM07R- and
// TSOPF-like matrices); otherwise the engine times both forms on the matrix and keeps the faster.  The overhang is read twice
// (by the finishing tile and by the next one), so the reach is kept small: <= 6% of a tile.  Measured with a reach of
// 2048 the TSOPF-like matrix (424 nnz/row) lost 15%: ~10% extra traffic plus a serial tail per block.
constexpr int kFlatFinish = 128;
void launch_break_points_from(hipStream_t stream, const int *rp, int m, int nnz, int stride, int tile0, int *bp, int bp_len); // bp[j] for absolute tile j + tile0
void launch_flat_digest(hipStream_t stream, const CsrDev &A, const FlatPlan &P); // after launch_break_points
void launch_flat_needs_fixup(hipStream_t stream, const CsrDev &A, const FlatPlan &P, int *d_flag); // d_flag[2] pre-zeroed
void launch_flat(hipStream_t stream, const CsrDev &A, const FlatPlan &P, double alpha, double beta, const double *x,
                 double *y);

// row-block-plus family: row blocks from the adaptive-plus analysis (break_points + first_block_of_row,
// made with threads_per_block = kPlusThreads).  launch_plus_digest (once per plan) packs one 16-B record per
// block into blk (nblocks * 16 bytes); the long-row fix-up runs only when the analysis found long rows.
// *d_has_long (pre-zeroed) is set to 1 if any block is a long-row slice.
void launch_plus_digest(hipStream_t stream, const CsrDev &A, const int *bp, const int *fbr, int nblocks, int long_chunk, void *blk,
                        int *d_has_long);
void launch_plus(hipStream_t stream, const CsrDev &A, const int *bp, const int *fbr, const void *blk, int nblocks,
                 bool has_long_rows, int xcd_chunk, int stream_policy, double *partial, double alpha, double beta,
                 const double *x, double *y, bool reverse = false);

// Device form of the row-block analysis (k_analyze.hip).  count: enqueue steps 1-3, d_total[0] = block count once
// the stream has run; emit: write break_points (total + 1 entries) and first_block_of_row (m + 1 entries).
size_t plus_analyze_device_workspace_bytes(int m);
bool plus_analyze_device_count(hipStream_t stream, const int *rp, int m, int min_nnz, int threads_per_block,
                               int vec_size, void *workspace, int *d_total);
void plus_analyze_device_emit(hipStream_t stream, const int *rp, int m, int min_nnz, const void *workspace, int *d_bp,
                              int *d_fbr);

// ---- opt-in column-slab blocking (k_slab.hip; tunable col_slabs) ---------------------------------------------------------------
// cnt / rps: S * (m + 1) ints, slab-major; off: S exclusive sums of the slabs' non-zero counts (device, 64-bit)
void launch_slab_count(hipStream_t stream, const CsrDev &A, int width, int S, int *cnt);
void launch_slab_scatter(hipStream_t stream, const CsrDev &A, int width, int S, const int *rps, const long long *off, int *ci_out,
                         double *v_out, bool values_only = false);
// compaction of one slab to its non-empty rows (flags -> caller's exclusive scan -> row ids + their row pointers) and the merge of a
// slab's compact result into y
void launch_slab_flags(hipStream_t stream, const int *rps, int m, int *flags);
void launch_slab_compact(hipStream_t stream, const int *rps, const int *pos, int m, int *rowid, int *crp);
void launch_slab_merge(hipStream_t stream, int ms, const int *rowid, const double *ys, double *y);
// value samples of a plan that holds a copy of the values: changed == nullptr records `count` evenly spaced samples of v[lo .. lo + span], else compares
// (saved: count 64-bit value patterns + count 32-bit column indices = 12 * count bytes; *changed: bit 0 values, bit 1 column indices)
void launch_value_samples(hipStream_t stream, const double *v, const int *ci, long long lo, long long span, int count, unsigned long long *saved, int *changed);
constexpr int kValueSamples = 65536;    // samples of the caller's values a plan with a copy of them re-checks before every use
constexpr int kSlabCopyAfterCalls = 32; // the automatic slab-major copy is built once a plan has served this many calls (or inside spmv_acc_prepare)

// dst = src over `bytes` (16-B granules) with the kernels' streaming load shape: the copy ceiling probe
void launch_stream_copy(hipStream_t stream, void *dst, const void *src, long long bytes, bool non_temporal);

// y[i] = beta * y[i] (used for m > 0, nnz == 0 and as a building block)
void launch_scale_y(hipStream_t stream, int m, double beta, double *y, const double *yin = nullptr);
void launch_validate_csr(hipStream_t stream, const CsrDev &A, int *d_flags); // *d_flags pre-zeroed; bits: see kernel

} // namespace spmv_acc
