// tuner.cpp -- what a plan is built from and what is timed on the matrix: the structural passes (break points, row-block analysis, digests,
// hints, slab lists) and the per-matrix timings with their budget (split out of engine.cpp in round 4; no behaviour change).
// Reference roles: hip-flat/flat.cpp:30-57 (break-point staging), hip-csr-adaptive-plus/csr_adaptive_plus_spmv.cpp:16-72 (analysis staging).
#include "engine_internal.hpp"

namespace spmv_acc {
using namespace detail;

namespace detail {

// Break points, carry buffers and the two plan-time probes of a flat plan with `stride` non-zeros per tile.
bool build_flat_plan(const CsrDev &A, int stride, hipStream_t stream, FlatPlan &F) {
  if (!plan_work_allowed("flat: break points and tile digests")) return false;
  ++t_plan_work;
  Plan::free_flat_plan(F);
  const int nnz = A.nnz;
  // (round 5) an un-rebased row sub-range (rowptr[0] = A.nnz0 > 0) starts at tile A.nnz0 / stride: the tiles before it own no rows and are
  // neither launched nor given table entries (chunk k of C used to launch up to C times the tiles it needed)
  const int tile0 = A.nnz0 / stride;
  const int tiles = nnz / stride + (nnz % stride ? 1 : 0) - tile0;
  const size_t n1 = static_cast<size_t>(tiles) + 1;
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&F.bp), sizeof(int) * n1), "hipMalloc break points") ||
      !hip_ok(hipMalloc(reinterpret_cast<void **>(&F.head), sizeof(double) * n1), "hipMalloc head carries") ||
      !hip_ok(hipMalloc(reinterpret_cast<void **>(&F.tail), sizeof(double) * n1), "hipMalloc tail carries") ||
      !hip_ok(hipMalloc(reinterpret_cast<void **>(&F.tail_row), sizeof(int) * n1), "hipMalloc tail rows") ||
      !hip_ok(hipMalloc(reinterpret_cast<void **>(&F.tail_end), sizeof(int) * n1), "hipMalloc tail ends") ||
      !hip_ok(hipMalloc(&F.digest, 16 * n1), "hipMalloc tile digest")) {
    Plan::free_flat_plan(F); // nothing half-built stays behind
    return false;
  }
  F.stride = stride;
  F.ntiles = tiles;
  F.tile0 = tile0;
  launch_break_points_from(stream, A.rp, A.m, nnz, stride, tile0, F.bp, static_cast<int>(n1));
  launch_flat_digest(stream, A, F);
  // does this matrix need the carry fix-up kernel at all? (only rows longer than a tile's finishing reach do)
  int *d_flag = nullptr;
  int h_flag[2] = {1, 0};
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d_flag), 2 * sizeof(int)), "hipMalloc flat flag")) {
    Plan::free_flat_plan(F);
    return false;
  }
  bool probed = hip_ok(hipMemsetAsync(d_flag, 0, 2 * sizeof(int), stream), "memset flat flag");
  if (probed) {
    launch_flat_needs_fixup(stream, A, F, d_flag);
    probed = hip_ok(hipMemcpyAsync(h_flag, d_flag, 2 * sizeof(int), hipMemcpyDeviceToHost, stream), "read flat flag") &&
             hip_ok(hipStreamSynchronize(stream), "sync flat flag");
  }
  (void)hipFree(d_flag);
  if (!probed) { // no probe result, an error is recorded: do not compute on guesses
    Plan::free_flat_plan(F);
    return false;
  }
  F.max_tile_rows = h_flag[1];
  F.can_finish = h_flag[0] == 0;
  F.needs_fixup = true; // until run_flat has timed both forms on this matrix
  F.mode_tuned[0] = F.mode_tuned[1] = false;
  return true;
}

// Matrices below this many non-zeros launch grids of a few workgroups per CU, where a tile kernel's chain of round trips is
// not hidden by other workgroups: there the tile size and the stream-first staging are timed per matrix as well.

thread_local bool t_flat_segment_sum = false; // this thread is inside segment_sum_flat_sparse_spmv (FlatSegmentSumScope)

int flat_stride_for(const Plan &p) {
  if (flat_segment_sum()) return kThreads * kNnzPerThread; // the scan is written for the 2048-non-zero tile
  if (tun(kT_col16) > 0) return kThreads * kNnzPerThread; // the 16-bit encoding (forced) is read by the 2048-non-zero tile
  int npt = tun(kT_flat_npt);
  if (npt < 0) npt = p.flat_npt_choice > 0 ? p.flat_npt_choice : kNnzPerThread;
  return kThreads * ((npt == 4 || npt == 16) ? npt : kNnzPerThread);
}

bool ensure_flat(Plan &p, hipStream_t stream) {
  const int stride = flat_stride_for(p);
  if (p.flat_tiles >= 0 && p.flat.stride == stride) return true;
  p.flat_tiles = -1;
  if (!build_flat_plan(p.A, stride, stream, p.flat)) return false;
  p.flat_tiles = p.flat.ntiles;
  // choices this matrix already has (timed earlier on a plan of another tile size, or loaded from the tune cache)
  if (p.flat_geometry_tuned) p.flat.early_stream = p.flat_early_choice && stride != kThreads * 16;
  for (int c = 0; c < 2; ++c) {
    if (p.flat_mode_choice[c] < 0) continue;
    p.flat.mode_tuned[c] = true;
    p.flat.tuned_fixup[c] = p.flat_mode_choice[c] == 1 || !p.flat.can_finish;
  }
  return true;
}

// Device form of the analysis into freshly allocated tables.  Returns the block count, or -1.
int analyze_on_device(hipStream_t stream, const int *d_rp, int m, int min_nnz, int threads, int vec, int **d_bp_out,
                      int **d_fbr_out) {
  void *ws = nullptr;
  int *d_total = nullptr;
  int blocks = -1;
  *d_bp_out = *d_fbr_out = nullptr;
  if (!hip_ok(hipMalloc(&ws, plus_analyze_device_workspace_bytes(m)), "hipMalloc analysis workspace")) return -1;
  if (hip_ok(hipMalloc(reinterpret_cast<void **>(&d_total), sizeof(int)), "hipMalloc analysis total")) {
    if (!plus_analyze_device_count(stream, d_rp, m, min_nnz, threads, vec, ws, d_total)) {
      set_error(kErrHip, "device row-block analysis: scan failed");
    } else {
      int total = 0;
      if (hip_ok(hipMemcpyAsync(&total, d_total, sizeof(int), hipMemcpyDeviceToHost, stream), "read block count") &&
          hip_ok(hipStreamSynchronize(stream), "sync analysis") &&
          hip_ok(hipMalloc(reinterpret_cast<void **>(d_bp_out), sizeof(int) * (static_cast<size_t>(total) + 1)),
                 "hipMalloc plus bp") &&
          hip_ok(hipMalloc(reinterpret_cast<void **>(d_fbr_out), sizeof(int) * (static_cast<size_t>(m) + 1)),
                 "hipMalloc plus fbr")) {
        plus_analyze_device_emit(stream, d_rp, m, min_nnz, ws, *d_bp_out, *d_fbr_out);
        if (hip_ok(hipStreamSynchronize(stream), "sync analysis emit")) blocks = total; // ws is freed below
      }
    }
    (void)hipFree(d_total);
  }
  (void)hipFree(ws);
  if (blocks < 0) {
    if (*d_bp_out) (void)hipFree(*d_bp_out);
    if (*d_fbr_out) (void)hipFree(*d_fbr_out);
    *d_bp_out = *d_fbr_out = nullptr;
  }
  return blocks;
}

} // namespace detail

int plus_analyze_device(int m, int min_nnz, int threads, int vec, const int *d_rowptr, int *d_bp, int bp_cap,
                        int *d_fbr) {
  int *tbp = nullptr, *tfbr = nullptr;
  hipStream_t st = get_stream();
  const int blocks = analyze_on_device(st, d_rowptr, m, min_nnz, threads, vec, &tbp, &tfbr);
  if (blocks < 0) return -2;
  int rc = blocks;
  if (blocks + 1 > bp_cap) {
    rc = -1;
  } else if (!hip_ok(hipMemcpyAsync(d_bp, tbp, sizeof(int) * (static_cast<size_t>(blocks) + 1), hipMemcpyDeviceToDevice, st),
                     "copy bp") ||
             !hip_ok(hipMemcpyAsync(d_fbr, tfbr, sizeof(int) * (static_cast<size_t>(m) + 1), hipMemcpyDeviceToDevice, st),
                     "copy fbr") ||
             !hip_ok(hipStreamSynchronize(st), "sync")) {
    rc = -2;
  }
  (void)hipFree(tbp);
  (void)hipFree(tfbr);
  return rc;
}

namespace detail {

bool ensure_plus(Plan &p, const int *h_rowptr, hipStream_t stream, int min_nnz) {
  if (min_nnz < 256 || min_nnz > kTile) min_nnz = kPlusMinNnz;
  const int want_vec =
      plus_pick_vec_tuned(p.A.m, p.A.count(), min_nnz);
  if (p.plus_blocks >= 0 && p.plus_vec == want_vec && p.plus_min == min_nnz) return true;
  if (!plan_work_allowed("row-block analysis")) return false;
  ++t_plan_work;
  auto drop_tables = [&p] { // also the exit of every failure below: nothing half-built stays behind
    if (p.d_pbp) (void)hipFree(p.d_pbp);
    if (p.d_pfbr) (void)hipFree(p.d_pfbr);
    if (p.d_ppartial) (void)hipFree(p.d_ppartial);
    if (p.d_pblk) (void)hipFree(p.d_pblk);
    p.d_pblk = nullptr;
    p.d_pbp = p.d_pfbr = nullptr;
    p.d_ppartial = nullptr;
    p.plus_blocks = -1;
    return false;
  };
  if (p.plus_blocks >= 0) (void)drop_tables(); // analysis parameters changed (measurement switch): rebuild
  // The reference picks VEC_SIZE = pow2 >= avg/2 (plus_pick_vec), which caps a block at THREADS/VEC rows and closes
  // most blocks far below MIN_NNZ_PER_BLOCK.  The analysis is the same function; only its row cap is chosen so
  // that cap * avg >= 1.25 * MIN_NNZ (blocks then close on their non-zero count).
  const int m = p.A.m;
  const int vec = want_vec;
  int blocks = -1;
  if (tun(kT_plus_host_analysis)) {
    // host form (the reference's): needs rowptr on the host
    std::vector<int> staged;
    const int *hrp = host_view(h_rowptr);
    if (!hrp) {
      staged.resize(static_cast<size_t>(m) + 1);
      if (!hip_ok(hipMemcpy(staged.data(), p.A.rp, sizeof(int) * (static_cast<size_t>(m) + 1), hipMemcpyDeviceToHost),
                  "stage rowptr for analysis"))
        return false;
      hrp = staged.data();
    }
    std::vector<int> bp, fbr;
    blocks = plus_analyze_host(m, min_nnz, kPlusThreads, vec, hrp, bp, fbr);
    // blocking copies: the host vectors die at scope exit (this runs once per matrix)
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_pbp), sizeof(int) * bp.size()), "hipMalloc plus bp") ||
        !hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_pfbr), sizeof(int) * fbr.size()), "hipMalloc plus fbr") ||
        !hip_ok(hipMemcpy(p.d_pbp, bp.data(), sizeof(int) * bp.size(), hipMemcpyHostToDevice), "copy plus bp") ||
        !hip_ok(hipMemcpy(p.d_pfbr, fbr.data(), sizeof(int) * fbr.size(), hipMemcpyHostToDevice), "copy plus fbr"))
      return drop_tables();
  } else {
    // device form: no host rowptr, no PCIe traffic beyond one int
    blocks = analyze_on_device(stream, p.A.rp, m, min_nnz, kPlusThreads, vec, &p.d_pbp, &p.d_pfbr);
    if (blocks < 0) return false;
  }
  int *d_flag = nullptr;
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_ppartial), sizeof(double) * (static_cast<size_t>(blocks) + 1)),
              "hipMalloc plus partial") ||
      !hip_ok(hipMalloc(&p.d_pblk, 16 * (static_cast<size_t>(blocks) + 1)), "hipMalloc plus digest") ||
      !hip_ok(hipMalloc(reinterpret_cast<void **>(&d_flag), sizeof(int)), "hipMalloc plus flag"))
    return drop_tables();
  int has_long = 0;
  bool ok = hip_ok(hipMemsetAsync(d_flag, 0, sizeof(int), stream), "memset plus flag");
  if (ok) {
    launch_plus_digest(stream, p.A, p.d_pbp, p.d_pfbr, blocks, 2 * min_nnz, p.d_pblk, d_flag);
    ok = hip_ok(hipMemcpyAsync(&has_long, d_flag, sizeof(int), hipMemcpyDeviceToHost, stream), "read plus flag") &&
         hip_ok(hipStreamSynchronize(stream), "sync plus digest");
  }
  (void)hipFree(d_flag);
  if (!ok) return drop_tables();
  p.plus_vec = vec;
  p.plus_min = min_nnz;
  p.plus_blocks = blocks;
  p.plus_has_long = has_long != 0;
  return true;
}

// beta class of the call being served (set by run_spmv): per-matrix timings run in the caller's class, into a zeroed scratch y
thread_local int t_beta_class = 1;

int policy_for(const Plan &p, int fam) {
  const int forced = tun(kT_stream_plain);
  if (forced >= 0) return forced & 3;
  // deterministic: the rule the timings follow on most matrices -- short rows (the vectors and rowptr are worth more cache than
  // the matrix) stream non-temporally, everything else with the default policy
  if (tun(kT_deterministic)) return static_cast<long long>(p.A.count()) <= 8LL * p.A.m ? kStreamPolicyNt : kStreamPolicyDefault;
  const int c = t_beta_class;
  if (p.stream_policy[fam][c] >= 0) return p.stream_policy[fam][c];
  // not timed for this family in this class yet (adaptive's comparison of the families): the policy another family measured on
  // this matrix in the same class is a far better guess than a fixed one, then this family's other class
  for (int f = 0; f < kFamilyCount; ++f)
    if (p.stream_policy[f][c] >= 0) return p.stream_policy[f][c];
  if (p.stream_policy[fam][c ^ 1] >= 0) return p.stream_policy[fam][c ^ 1];
  return static_cast<long long>(p.A.count()) <= 8LL * p.A.m ? kStreamPolicyNt : kStreamPolicyDefault; // (nothing measured yet: the rule)
}

// While adaptive compares the families it runs each with its default sub-choices (flat: carries + fix-up unless pinned;
// row-block-plus: MIN_NNZ 1536); the family that wins refines its own sub-choice on its next call.
// plan-time budget state (engine_internal.hpp: defer_tuning / by_rule)
thread_local std::chrono::steady_clock::time_point t_call_began;
thread_local double t_budget_spmvs = 0.0;
thread_local double t_budget_ms = -1.0;
thread_local bool t_tuning_deferred = false;
thread_local bool t_rule_twin = false;
thread_local float t_first_trial_ms = 0.f;
thread_local int t_unbounded_tuning = 0;
thread_local bool t_coarse_tuning = false;
thread_local bool t_no_policy_timing = false; // run_plus's early slab decision: the row-block-plus kernel runs under the rule's cache policy, nothing is timed for it


// SPMV_ACC_TUNE_LOG=1: every per-matrix timing and the choice it led to, one line each on stderr (what was measured, not only
// what was kept -- for users who want to pin a choice, and for finding out why a plan settled where it did).
bool tune_log_enabled() {
  static const bool on = [] {
    const char *e = std::getenv("SPMV_ACC_TUNE_LOG");
    return e && *e && *e != '0';
  }();
  return on;
}
void tune_log(const char *fmt, ...) {
  if (!tune_log_enabled()) return;
  va_list ap;
  va_start(ap, fmt);
  std::fputs("[spmv_acc tune] ", stderr);
  std::vfprintf(stderr, fmt, ap);
  std::fputc('\n', stderr);
  va_end(ap);
}

// Shared by the per-matrix timings below: average milliseconds of fn() in the cache state fn itself leaves behind.  The
// first launch is timed alone and sizes the rest, so tuning a matrix whose SpMV takes milliseconds costs 2 launches per
// candidate, not 8: under 0.1 ms per launch 2 more warm-ups + 5 timed, under 0.5 ms 1 + 3, under 2 ms 1 + 2, else the one warm-up + 1 timed.
// ONE scratch y per run_spmv call, shared by every timing phase of that call (cache policy, block sizes, hints, adaptive's families, the
// slab passes): each phase used to hipMalloc / hipFree its own -- a device synchronisation apiece, and seconds apiece at 2 G rows (17 GB).
// The phases nest (adaptive's family timing calls the families' own timings): the content is never read, only written and reset.
thread_local double *t_scratch = nullptr;
thread_local size_t t_scratch_len = 0;
double *tune_scratch(size_t len) {
  if (t_scratch && t_scratch_len >= len) return t_scratch;
  if (t_scratch) (void)hipFree(t_scratch);
  t_scratch = nullptr;
  t_scratch_len = 0;
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&t_scratch), sizeof(double) * (len ? len : 1)), "hipMalloc tune y")) {
    t_scratch = nullptr;
    return nullptr;
  }
  t_scratch_len = len;
  return t_scratch;
}
void release_tune_scratch() {
  if (t_scratch) (void)hipFree(t_scratch);
  t_scratch = nullptr;
  t_scratch_len = 0;
}



// The 16-bit column encoding of the view's non-zeros (k_col16.hip): base + escape count per 256-non-zero chunk (and the escape statistics
// that choose the record size), exclusive scan of the overflow counts, then offsets, records and overflow list.  Two synchronisations
// (the statistics choose R, the overflow total sizes the last allocation).  Leaves state 0 where the encoding cannot pay: x of 4 GB or
// more (the kernels' 32-bit gather offsets), fewer than 64 chunks, or more than 1 % of the chunks overflowing even the 64-int record.
bool ensure_col16(Plan &p, hipStream_t st) {
  Col16 &C = p.col16;
  if (C.state >= 0) return true;
  if (!plan_work_allowed("the 16-bit column encoding")) return false;
  const CsrDev &A = p.A;
  // (the first chunk: aligned down to a whole 2048-non-zero flat tile -- 8 chunks -- because a flat tile's origin is a multiple of 2048 and its first
  // wavefronts stage from there; the row blocks' tile origins are multiples of 256 at or above nnz0's chunk)
  const int chunk0 = A.nnz0 / (kThreads * kNnzPerThread) * (kThreads * kNnzPerThread / kCol16Chunk);
  const int nchunks = (A.nnz + kCol16Chunk - 1) / kCol16Chunk - chunk0;
  if (!x32_ok(A) || nchunks < 64) {
    C.state = 0;
    return true;
  }
  ++t_plan_work;
  int *base = nullptr, *cnt = nullptr, *ovf_start = nullptr;
  unsigned long long *stats = nullptr;
  void *tmp = nullptr;
  const size_t tmp_bytes = col16_scan_bytes(nchunks);
  const size_t n1 = static_cast<size_t>(nchunks) + 1;
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&base), sizeof(int) * n1), "hipMalloc col16 bases") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&cnt), sizeof(int) * n1), "hipMalloc col16 escape counts") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&ovf_start), sizeof(int) * n1), "hipMalloc col16 overflow offsets") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&stats), sizeof(unsigned long long) * 4), "hipMalloc col16 statistics") &&
            hip_ok(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16), "hipMalloc col16 scan workspace") &&
            hip_ok(hipMemsetAsync(cnt + nchunks, 0, sizeof(int), st), "memset col16") &&
            hip_ok(hipMemsetAsync(stats, 0, sizeof(unsigned long long) * 4, st), "memset col16 statistics");
  unsigned long long h_stats[4] = {0, 0, 0, 0};
  int R = 0, total = 0;
  if (ok) {
    launch_col16_base(st, A.ci, A.nnz, chunk0, nchunks, base, cnt, stats);
    ok = hip_ok(hipMemcpyAsync(h_stats, stats, sizeof(h_stats), hipMemcpyDeviceToHost, st), "read col16 statistics") &&
         hip_ok(hipStreamSynchronize(st), "sync col16 statistics");
  }
  if (ok) {
    const unsigned long long limit = static_cast<unsigned long long>(nchunks) / 100; // chunks allowed to overflow their record
    R = h_stats[1] <= limit ? 16 : (h_stats[2] <= limit ? 32 : (h_stats[3] <= limit ? 64 : 0));
    const int forced = tun(kT_col16) > 1 ? tun(kT_col16) : 0; // (tests: col16 = 16 / 32 / 64 pins the record size, overflow or not)
    if (forced == 16 || forced == 32 || forced == 64) R = forced;
    tune_log("m %d nnz %d 16-bit columns: %llu escapes in %d chunks, %llu / %llu / %llu chunks above 12 / 28 / 60 -> %s", A.m, A.nnz, h_stats[0], nchunks,
             h_stats[1], h_stats[2], h_stats[3], R ? (R == 16 ? "16-int records" : (R == 32 ? "32-int records" : "64-int records")) : "not encoded");
  }
  if (ok && R > 0) {
    launch_col16_overflow(st, cnt, nchunks, R - 4);
    ok = launch_col16_scan(st, nchunks, cnt, ovf_start, tmp, tmp_bytes) &&
         hip_ok(hipMemcpyAsync(&total, ovf_start + nchunks, sizeof(int), hipMemcpyDeviceToHost, st), "read col16 overflow total") &&
         hip_ok(hipStreamSynchronize(st), "sync col16");
    const size_t d16_bytes = sizeof(unsigned short) * static_cast<size_t>(nchunks) * kCol16Chunk;
    const size_t rec_bytes = sizeof(int) * static_cast<size_t>(nchunks) * R;
    const size_t ovf_bytes = sizeof(int) * (static_cast<size_t>(total) + 64);
    ok = ok && hip_ok(hipMalloc(reinterpret_cast<void **>(&C.d16), d16_bytes), "hipMalloc col16 offsets") &&
         hip_ok(hipMalloc(reinterpret_cast<void **>(&C.rec), rec_bytes), "hipMalloc col16 records") &&
         hip_ok(hipMalloc(reinterpret_cast<void **>(&C.ovf), ovf_bytes), "hipMalloc col16 overflow list") &&
         hip_ok(hipMalloc(reinterpret_cast<void **>(&C.ci_guard), sizeof(int) * kGuardSamples), "hipMalloc col16 guard") &&
         hip_ok(hipMemsetAsync(C.rec, 0, rec_bytes, st), "memset col16 records") && hip_ok(hipMemsetAsync(C.ovf, 0, ovf_bytes, st), "memset col16 overflow");
    if (ok) {
      launch_col16_encode(st, A.ci, A.nnz, chunk0, nchunks, base, ovf_start, R, C.d16, C.rec, C.ovf);
      launch_col16_guard(st, A.ci, A.nnz0, A.count() > 0 ? A.count() - 1 : 0, C.ci_guard);
      ok = hip_ok(hipStreamSynchronize(st), "sync col16 encode");
    }
    C.bytes = d16_bytes + rec_bytes + ovf_bytes;
  }
  for (void *q : {static_cast<void *>(base), static_cast<void *>(cnt), static_cast<void *>(ovf_start), static_cast<void *>(stats), tmp})
    if (q) (void)hipFree(q);
  if (!ok) {
    p.free_col16();
    return false;
  }
  if (R == 0) {
    p.free_col16();
    C.state = 0;
    return true;
  }
  C.state = 1;
  C.chunk0 = chunk0;
  C.nchunks = nchunks;
  C.rec_ints = R;
  C.escapes = static_cast<long long>(h_stats[0]);
  C.overflow = total;
  return true;
}

// walking direction of this plan's next tile-kernel launch (tunable zigzag): consecutive SpMVs on a matrix alternate

void launch_flat_plan(hipStream_t st, const CsrDev &A, FlatPlan &F, int policy, double alpha, double beta, const double *x,
                      double *y, bool reverse) {
  F.xcd_chunk = tun(kT_xcd_chunk);
  F.stream_policy = policy;
  const int early = tun(kT_flat_early);
  if (early >= 0) F.early_stream = early != 0; // pinned (A/B runs); otherwise the plan's timed choice
  F.reverse = reverse;
  F.cache_ends = tun(kT_cache_ends_mb) > 0 && tun(kT_zigzag) ? static_cast<int>(tun(kT_cache_ends_mb) * 1048576.0 / (12.0 * F.stride)) : 0;
  launch_flat(st, A, F, alpha, beta, x, y);
}
void launch_flat_with(hipStream_t st, Plan &p, int policy, double alpha, double beta, const double *x, double *y) {
  p.flat.segment_sum = flat_segment_sum();
  launch_flat_plan(st, p.A, p.flat, policy, alpha, beta, x, y, next_reverse(p));
}

// Cut rows of a flat plan without long overhangs can be folded two ways (kernels.hpp kFlatFinish).  Which is faster
// depends on the matrix: the finishing wave lengthens every workgroup by a dependent global load (-5 % on 150 us
// kernels) but saves the fix-up launch (+5..20 % on kernels under 30 us).  Timed once per matrix like the cache policy.
bool autotune_flat_mode(Plan &p, hipStream_t st, const double *x) {
  FlatPlan &F = p.flat;
  const int forced = tun(kT_flat_finish);
  if (!F.can_finish || F.ntiles <= 1) {
    F.needs_fixup = !F.can_finish;
    return true;
  }
  if (forced >= 0) { // pinned (A/B runs): follows the tunable on every call
    F.needs_fixup = forced == 0;
    return true;
  }
  if (tun(kT_deterministic)) { // by rule: tiles finish their cut rows wherever that is legal (what the timing picks on most matrices)
    F.needs_fixup = false;
    return true;
  }
  const int cls = t_beta_class; // (the fix-up kernel re-reads the old y of every cut row: the two forms rank per beta class)
  if (F.mode_tuned[cls]) {
    F.needs_fixup = F.tuned_fixup[cls];
    return true;
  }
  if (!t_capturing && defer_tuning()) { // the call's tuning budget is spent: the rule for now, timed by a later call
    F.needs_fixup = false;
    return true;
  }
  if (t_capturing) { // not timed in this class yet and no timing inside a capture: the other class' choice, else finish in the tile
    F.needs_fixup = F.mode_tuned[cls ^ 1] ? F.tuned_fixup[cls ^ 1] : false;
    return true;
  }
  if (t_coarse_tuning && p.A.count() >= flat_small_nnz()) { // (small matrices: the timings are cheap and decide the comparison)
    F.needs_fixup = true;
    return true;
  }
  ++t_plan_work;
  double *scratch = nullptr;
  if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
  TuneTimer timer;
  timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
  bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
  float ms[2] = {0.f, 0.f};
  for (int mode = 0; ok && mode < 2; ++mode) {
    F.needs_fixup = mode == 0;
    ok = timer.time(st, [&] { launch_flat_with(st, p, policy_for(p, kFamFlat), 1.0, trial_beta(), x, scratch); }, &ms[mode]);
  }
  F.tuned_fixup[cls] = F.needs_fixup = !(ok && ms[1] < ms[0]);
  F.mode_tuned[cls] = ok;
  if (ok) p.flat_mode_choice[cls] = F.tuned_fixup[cls] ? 1 : 0;
  if (ok) tune_log("m %d nnz %d beta class %d flat cut rows: carries + fix-up %.2f us, finished in the tile %.2f us", p.A.m, p.A.nnz, cls, ms[0] * 1e3f, ms[1] * 1e3f);
  return ok;
}

// Small grids (kFlatSmallNnz): time {this tile size, the other one} x {stream loads first, break-point chain first} once
// per matrix and keep the fastest.  The other tile size gets its own break points / carries; its cut rows are finished in
// the tile whenever that is legal (no second launch: what wins on short kernels).
bool autotune_flat_geometry(Plan &p, hipStream_t st, const double *x) {
  if (p.flat_geometry_tuned || tun(kT_col16) > 0 || flat_segment_sum() || t_capturing || by_rule()) return true;
  if (p.A.count() >= flat_small_nnz() || p.flat.ntiles <= 1) {
    p.flat_geometry_tuned = true;
    return true;
  }
  const bool time_npt = tun(kT_flat_npt) < 0, time_early = tun(kT_flat_early) < 0;
  if (!time_npt && !time_early) return true; // pinned (A/B runs)
  ++t_plan_work;
  double *scratch = nullptr;
  if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
  TuneTimer timer;
  timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
  bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
  const int pol = policy_for(p, kFamFlat);
  FlatPlan alt;
  FlatPlan *plans[2] = {&p.flat, nullptr};
  if (ok && time_npt) {
    const int other = p.flat.stride == kThreads * 4 ? kThreads * kNnzPerThread : kThreads * 4;
    ok = build_flat_plan(p.A, other, st, alt);
    if (ok) {
      alt.needs_fixup = alt.tuned_fixup[0] = alt.tuned_fixup[1] = !alt.can_finish;
      alt.mode_tuned[0] = alt.mode_tuned[1] = true;
      plans[1] = &alt;
    }
  }
  float best = 1e30f;
  int best_plan = 0;
  bool best_early = false;
  for (int k = 0; ok && k < 2; ++k) {
    if (!plans[k]) continue;
    for (int e = 0; ok && e < (time_early ? 2 : 1); ++e) {
      plans[k]->early_stream = time_early ? e != 0 : plans[k]->early_stream;
      float ms = 0.f;
      ok = timer.time(st, [&] { launch_flat_plan(st, p.A, *plans[k], pol, 1.0, trial_beta(), x, scratch, next_reverse(p)); }, &ms);
      if (ok) tune_log("m %d nnz %d flat geometry: %d non-zeros per tile, stream-first %d -> %.2f us", p.A.m, p.A.nnz, plans[k]->stride,
                       plans[k]->early_stream ? 1 : 0, ms * 1e3f);
      if (ok && ms < best) {
        best = ms;
        best_plan = k;
        best_early = plans[k]->early_stream;
      }
    }
  }
  if (ok) {
    if (best_plan == 1) {
      Plan::free_flat_plan(p.flat);
      p.flat = alt;
      alt = FlatPlan(); // ownership moved
      p.flat_tiles = p.flat.ntiles;
    }
    p.flat.early_stream = p.flat_early_choice = best_early;
    p.flat_npt_choice = p.flat.stride / kThreads;
    p.flat_geometry_tuned = true;
    for (int c = 0; c < 2; ++c) p.flat_mode_choice[c] = p.flat.mode_tuned[c] ? (p.flat.tuned_fixup[c] ? 1 : 0) : -1;
  }
  Plan::free_flat_plan(alt);
  return ok;
}

// Gather hints (k_hint.hip): census of the matrix' columns -> hot set of x lines within the budget -> one cold bit per non-zero.
// Returns false on a HIP failure only; p.hint_state says whether hints exist.
bool ensure_hint(Plan &p, hipStream_t st) {
  if (p.hint_state >= 0) return true;
  const int mode = tun(kT_gather_hint);
  p.hint_state = 0;
  const CsrDev &A = p.A;
  // hinted gathers address x by 32-bit byte offsets; a matrix whose x fits an L2 several times over has nothing to protect
  if (mode == 0 || A.count() < 8 || A.n <= 0 || static_cast<long long>(A.n) * 8 >= (1LL << 32)) return true;
  // (and while x lives in the 256 MB Infinity Cache beside the rest of the working set a cold gather is a hit there, which a non-temporal
  // load forfeits: R-MAT scale 21 / 22 / 23, x = 16 / 32 / 64 MB: hinted 268 / 604 / 1343 us against 212 / 477 / 1250 plain; scale 24 / 25,
  // x = 128 / 256 MB: 3.03 / 7.2 ms against 3.38 / 8.2 -- the timed choice gets all five right, this bound just saves the census)
  if (mode < 0 && static_cast<long long>(A.n) * 8 < (static_cast<long long>(tun(kT_hint_min_x_mb)) << 20)) return true;
  ++t_plan_work;
  const auto census_t0 = std::chrono::steady_clock::now();
  const int nlines = (A.n + (1 << kHintLineShift) - 1) >> kHintLineShift;
  // (odd: an even stride on rows of one even length would sample the same position of every row -- with sorted rows always low columns)
  // (round 5: the census and the bits cover the view's own non-zeros [A.nnz0, A.nnz); the bitmap stays indexed by absolute position)
  const int stride = ((A.count() + kHintSamples - 1) / kHintSamples) | 1;
  const int samples = (A.count() + stride - 1) / stride;
  unsigned *counts = nullptr, *hist_lines = nullptr;
  unsigned long long *hist_hits = nullptr;
  std::vector<unsigned> h_lines(kHintBins);
  std::vector<unsigned long long> h_hits(kHintBins);
  // (hints are optional: a matrix that fills the card leaves no room for them, and that must not fail the SpMV)
  auto optional_alloc = [](void **ptr, size_t bytes) {
    if (hipMalloc(ptr, bytes) == hipSuccess) return true;
    (void)hipGetLastError(); // clear the sticky out-of-memory error
    *ptr = nullptr;
    return false;
  };
  if (!optional_alloc(reinterpret_cast<void **>(&counts), sizeof(unsigned) * static_cast<size_t>(nlines)) ||
      !optional_alloc(reinterpret_cast<void **>(&hist_lines), sizeof(unsigned) * kHintBins) ||
      !optional_alloc(reinterpret_cast<void **>(&hist_hits), sizeof(unsigned long long) * kHintBins)) {
    if (counts) (void)hipFree(counts);
    if (hist_lines) (void)hipFree(hist_lines);
    if (hist_hits) (void)hipFree(hist_hits);
    return true;
  }
  bool ok = hip_ok(hipMemsetAsync(counts, 0, sizeof(unsigned) * static_cast<size_t>(nlines), st), "memset hint census") &&
            hip_ok(hipMemsetAsync(hist_lines, 0, sizeof(unsigned) * kHintBins, st), "memset hint histogram") &&
            hip_ok(hipMemsetAsync(hist_hits, 0, sizeof(unsigned long long) * kHintBins, st), "memset hint histogram");
  if (ok) {
    launch_hint_census(st, A.ci + A.nnz0, A.count(), A.n, stride, samples, counts);
    launch_hint_hist(st, counts, nlines, hist_lines, hist_hits);
    ok = hip_ok(hipMemcpyAsync(h_lines.data(), hist_lines, sizeof(unsigned) * kHintBins, hipMemcpyDeviceToHost, st), "read hint histogram") &&
         hip_ok(hipMemcpyAsync(h_hits.data(), hist_hits, sizeof(unsigned long long) * kHintBins, hipMemcpyDeviceToHost, st), "read hint histogram") &&
         hip_ok(hipStreamSynchronize(st), "sync hint census");
  }
  if (ok) {
    // hot set = the lines with the highest counts that fit the budget; T = the smallest count still inside it
    const long long budget_lines = static_cast<long long>(tun(kT_hint_budget_kb) > 0 ? tun(kT_hint_budget_kb) : 1) * 1024 / (8 << kHintLineShift);
    unsigned long long total = 0, hot = 0;
    long long lines = 0, touched = 0;
    for (int b = 1; b < kHintBins; ++b) {
      total += h_hits[b];
      touched += h_lines[b];
    }
    unsigned threshold = kHintBins; // nothing hot
    for (int b = kHintBins - 1; b >= 1; --b) {
      if (lines + h_lines[b] > budget_lines) break;
      lines += h_lines[b];
      hot += h_hits[b];
      threshold = static_cast<unsigned>(b);
    }
    p.hint_hot_share = total ? static_cast<double>(hot) / static_cast<double>(total) : 0.0;
    // worth a timed look (bits pass + two timings): the hot set takes a real share of the gathers, its lines are at least three times
    // as popular as the average touched line (FEM-like matrices: every line is touched about equally often, ratio ~1; R-MAT 25: 12),
    // and enough cold gathers exist to do the displacing
    const bool candidate = lines > 0 && p.hint_hot_share >= 0.15 && p.hint_hot_share <= 0.95 &&
                           p.hint_hot_share * static_cast<double>(touched) >= 3.0 * static_cast<double>(lines);
    tune_log("m %d nnz %d column census (%.2f ms): %d samples, %lld of %lld touched x lines hot (count >= %u), %.1f %% of the gathers%s", A.m, A.nnz,
             std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - census_t0).count(), samples, lines, touched, threshold,
             100.0 * p.hint_hot_share, candidate || mode > 0 ? "" : " -> no hints");
    if (candidate || mode > 0) {
      const size_t nbytes = (static_cast<size_t>(A.nnz) + 7) / 8 + 16;
      const bool room = optional_alloc(reinterpret_cast<void **>(&p.d_cold), nbytes);
      ok = !room || hip_ok(hipMemsetAsync(p.d_cold, 0, nbytes, st), "memset hint bits");
      if (ok && room) {
        const int skip = A.nnz0 & ~7; // whole bitmap bytes before the view
        launch_hint_bits(st, A.ci + skip, A.nnz - skip, A.n, counts, threshold, p.d_cold + skip / 8);
        ok = hip_ok(hipStreamSynchronize(st), "sync hint bits");
      }
      if (ok && room) p.hint_state = 1;
      else if (p.d_cold) {
        (void)hipFree(p.d_cold);
        p.d_cold = nullptr;
      }
    }
  }
  if (counts) (void)hipFree(counts);
  if (hist_lines) (void)hipFree(hist_lines);
  if (hist_hits) (void)hipFree(hist_hits);
  return ok;
}


bool probe_rowblock(Plan &p, int rpb, hipStream_t st);

} // namespace detail

namespace detail {

// Opt-in column-slab blocking: build the S slabs of this matrix (k_slab.hip) once per plan and S.
bool ensure_slabs(Plan &p, int S, hipStream_t st) {
  if (p.d_slab_rp && p.slab_count == S) return true;
  if (!plan_work_allowed("building the column slabs")) return false;
  ++t_plan_work;
  p.free_slabs();
  const CsrDev &A = p.A;
  const size_t m1 = static_cast<size_t>(A.m) + 1;
  const int width = (A.n + S - 1) / S > 0 ? (A.n + S - 1) / S : 1;
  int *cnt = nullptr;
  long long *d_off = nullptr;
  void *tmp = nullptr;
  const size_t tmp_bytes = col16_scan_bytes(A.m);
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&cnt), sizeof(int) * m1 * S), "hipMalloc slab counts") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_slab_rp), sizeof(int) * m1 * S), "hipMalloc slab rowptr") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_slab_ci), sizeof(int) * (static_cast<size_t>(A.nnz) + 4)), "hipMalloc slab colindex") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_slab_v), sizeof(double) * (static_cast<size_t>(A.nnz) + 4)), "hipMalloc slab values") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&d_off), sizeof(long long) * S), "hipMalloc slab offsets") &&
            hip_ok(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16), "hipMalloc slab scan workspace");
  // (the re-ordered colindex starts as zeros: whatever a consumer reads beyond the slabs' non-zeros is a valid column)
  ok = ok && hip_ok(hipMemsetAsync(p.d_slab_ci, 0, sizeof(int) * (static_cast<size_t>(A.nnz) + 4), st), "memset slab colindex");
  std::vector<long long> off(S, 0);
  long long slab_total = 0; // (a row shard handed over without rebasing: fewer than A.nnz, which is then the END offset)
  if (ok) {
    launch_slab_count(st, A, width, S, cnt);
    for (int s = 0; ok && s < S; ++s) // exclusive scan over m + 1 entries: rowptr_s, with rowptr_s[m] = the slab's non-zero count
      ok = launch_col16_scan(st, A.m, cnt + m1 * s, p.d_slab_rp + m1 * s, tmp, tmp_bytes);
    if (!ok) set_error(kErrHip, "column slabs: scan failed");
    long long run = 0;
    for (int s = 0; ok && s < S; ++s) {
      int total = 0;
      ok = hip_ok(hipMemcpyAsync(&total, p.d_slab_rp + m1 * s + A.m, sizeof(int), hipMemcpyDeviceToHost, st), "read slab size") &&
           hip_ok(hipStreamSynchronize(st), "sync slab size");
      off[s] = run;
      run += total;
    }
    slab_total = run;
    ok = ok && hip_ok(hipMemcpyAsync(d_off, off.data(), sizeof(long long) * S, hipMemcpyHostToDevice, st), "write slab offsets");
    if (ok) {
      launch_slab_scatter(st, A, width, S, p.d_slab_rp, d_off, p.d_slab_ci, p.d_slab_v);
      ok = hip_ok(hipStreamSynchronize(st), "sync slab scatter");
    }
    // compact every slab to its non-empty rows (cnt is free now: its first two (m + 1)-blocks serve as flags and positions)
    p.slab_rowid.assign(S, nullptr);
    p.slab_crp.assign(S, nullptr);
    p.slab_rows.assign(S, 0);
    int *flags = cnt, *pos = S >= 2 ? cnt + m1 : nullptr;
    int max_rows = 0;
    for (int s = 0; ok && s < S && pos; ++s) {
      const int *rps = p.d_slab_rp + m1 * s;
      launch_slab_flags(st, rps, A.m, flags);
      int ms = 0;
      ok = launch_col16_scan(st, A.m, flags, pos, tmp, tmp_bytes) &&
           hip_ok(hipMemcpyAsync(&ms, pos + A.m, sizeof(int), hipMemcpyDeviceToHost, st), "read slab rows") &&
           hip_ok(hipStreamSynchronize(st), "sync slab rows") &&
           hip_ok(hipMalloc(reinterpret_cast<void **>(&p.slab_rowid[s]), sizeof(int) * (static_cast<size_t>(ms) + 1)), "hipMalloc slab row ids") &&
           hip_ok(hipMalloc(reinterpret_cast<void **>(&p.slab_crp[s]), sizeof(int) * (static_cast<size_t>(ms) + 1)), "hipMalloc slab compact rowptr");
      if (!ok) break;
      launch_slab_compact(st, rps, pos, A.m, p.slab_rowid[s], p.slab_crp[s]);
      ok = hip_ok(hipStreamSynchronize(st), "sync slab compaction"); // (flags / pos are reused by the next slab)
      p.slab_rows[s] = ms;
      max_rows = ms > max_rows ? ms : max_rows;
    }
    ok = ok && hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_slab_ys), sizeof(double) * (static_cast<size_t>(max_rows) + 1)), "hipMalloc slab result");
  }
  if (cnt) (void)hipFree(cnt);
  if (tmp) (void)hipFree(tmp);
  if (!ok) {
    if (d_off) (void)hipFree(d_off);
    p.free_slabs();
    return false;
  }
  p.d_slab_off = d_off;
  p.slab_width = width;
  p.slab_count = S;
  p.slab_off = off;
  p.slab_off.push_back(slab_total);
  return true;
}

// Column-slab blocking without a copy (tunable slab_segments): the per-slab run lists of k_segment.hip.  Structure only; built once.
bool ensure_segments(Plan &p, int S_cols, hipStream_t st) {
  // two-class form (tunable slab_whole_below): only the rows of at least that many non-zeros are cut by column slab; every shorter row is ONE run,
  // all columns, in a pass of its own (plane S_cols)
  const int rest_below = tun(kT_slab_whole_below) > 1 ? tun(kT_slab_whole_below) : 0;
  // (kSegMaxPlanes planes in all: the count kernels keep one counter per plane in 16 lanes / 16 packed bytes.  The automatic slab count reaches
  // 16 from x = 496 MB on, and until this clamp the whole-row plane was then a 17th: rows of exactly 32 non-zeros -- the only ones the one-lane count
  // kernel cuts by slab -- had slab 8's run filed twice.  Found by profiles/probes/rmat26_check.py, 130,277 wrong rows on R-MAT 26)
  if (S_cols > kSegMaxPlanes - (rest_below > 0 ? 1 : 0)) S_cols = kSegMaxPlanes - (rest_below > 0 ? 1 : 0);
  const int S = S_cols + (rest_below > 0 ? 1 : 0); // planes
  if (p.seg_state >= 0 && (p.seg_state == 0 || (p.seg_slabs == S && p.seg_rest_below == rest_below))) return true;
  if (!plan_work_allowed("building the column-slab run lists")) return false;
  ++t_plan_work;
  p.free_segments();
  const CsrDev &A = p.A;
  const size_t m1 = static_cast<size_t>(A.m) + 1;
  const int width = (A.n + S_cols - 1) / S_cols > 0 ? (A.n + S_cols - 1) / S_cols : 1;
  SlabBounds bounds;
  for (int b = 0; b < 15; ++b) bounds.first[b] = static_cast<int>(std::min<long long>(static_cast<long long>(width) * (b + 1), INT_MAX));
  // (equal column ranges.  Unequal ones were tried on R-MAT 25 through an environment hook since removed -- the hot eighth split in two
  // or four, the cold half kept whole, 4 to 8 slabs in all: 5.53-5.92 ms against 5.30 for eight equal slabs, profiles/r03_slab_segments.txt)
  int *cnt = nullptr, *beg = nullptr, *pieces = nullptr, *pos = nullptr, *flag = nullptr;
  void *tmp = nullptr;
  const size_t tmp_bytes = col16_scan_bytes(A.m);
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&cnt), sizeof(int) * m1 * S), "hipMalloc run counts") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&beg), sizeof(int) * m1 * S), "hipMalloc run starts") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&pieces), sizeof(int) * m1), "hipMalloc run pieces") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&pos), sizeof(int) * m1), "hipMalloc run positions") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&flag), sizeof(int)), "hipMalloc order flag") &&
            hip_ok(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16), "hipMalloc scan workspace") &&
            hip_ok(hipMemsetAsync(flag, 0, sizeof(int), st), "memset order flag");
  int unordered = 0;
  if (ok) {
    ok = launch_segment_count(st, A, bounds, S, cnt, beg, flag, rest_below);
    if (!ok) set_error(kErrBadArgument, "slab run lists: more planes than the count kernels hold");
    ok = ok && hip_ok(hipMemcpyAsync(&unordered, flag, sizeof(int), hipMemcpyDeviceToHost, st), "read order flag") &&
         hip_ok(hipStreamSynchronize(st), "sync run counts");
  }
  if (ok && unordered) {
    p.seg_state = 0; // some row's columns do not ascend across a slab boundary: its slab parts are not runs
    tune_log("m %d nnz %d: slab_segments: rows are not ordered by column slab, ordinary path", A.m, A.nnz);
  } else if (ok) {
    for (auto *list : {&p.seg_row, &p.seg_begin, &p.seg_vptr, &p.seg_blk, &p.seg_cut}) list->assign(S, nullptr);
    p.seg_entries.assign(S, 0);
    p.seg_blocks.assign(S, 0);
    p.seg_pieces.assign(S, 0);
    int max_entries = 0;
    for (int s = 0; ok && s < S; ++s) {
      const int *cnt_s = cnt + m1 * s;
      int entries = 0;
      launch_segment_pieces(st, cnt_s, A.m, kSegPiece, pieces);
      ok = launch_col16_scan(st, A.m, pieces, pos, tmp, tmp_bytes) &&
           hip_ok(hipMemcpyAsync(&entries, pos + A.m, sizeof(int), hipMemcpyDeviceToHost, st), "read run count") &&
           hip_ok(hipStreamSynchronize(st), "sync run count");
      if (!ok || entries == 0) continue;
      // (the cost prefix is an int scan: a pass of hundreds of millions of one-element runs would overflow it -- such a matrix
      // keeps the ordinary path)
      if (static_cast<long long>(entries) * 4 + A.nnz > static_cast<long long>(INT_MAX) - 65536) {
        unordered = 2;
        break;
      }
      // entries: row, first non-zero, length -> vptr; cost -> cptr -> the workgroups' first entries
      const size_t e1 = static_cast<size_t>(entries) + 1;
      int *len = nullptr, *cost = nullptr, *cptr = nullptr;
      void *tmp_e = nullptr;
      const size_t tmp_e_bytes = col16_scan_bytes(entries);
      long long total_cost = 0;
      int last[2] = {0, 0};
      ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&p.seg_row[s]), sizeof(int) * e1), "hipMalloc run rows") &&
           hip_ok(hipMalloc(reinterpret_cast<void **>(&p.seg_begin[s]), sizeof(int) * e1), "hipMalloc run starts") &&
           hip_ok(hipMalloc(reinterpret_cast<void **>(&p.seg_vptr[s]), sizeof(int) * e1), "hipMalloc run prefix") &&
           hip_ok(hipMalloc(reinterpret_cast<void **>(&len), sizeof(int) * e1), "hipMalloc run lengths") &&
           hip_ok(hipMalloc(reinterpret_cast<void **>(&cost), sizeof(int) * e1), "hipMalloc run costs") &&
           hip_ok(hipMalloc(reinterpret_cast<void **>(&cptr), sizeof(int) * e1), "hipMalloc run cost prefix") &&
           hip_ok(hipMalloc(&tmp_e, tmp_e_bytes > 0 ? tmp_e_bytes : 16), "hipMalloc scan workspace");
      if (ok) {
        ok = hip_ok(hipMemsetAsync(flag, 0, sizeof(int), st), "memset piece flag");
        launch_segment_compact(st, cnt_s, beg + m1 * s, pos, A.m, kSegPiece, p.seg_row[s], p.seg_begin[s], len, flag);
        launch_segment_cost(st, entries, len, cost);
        ok = ok && launch_col16_scan(st, entries, len, p.seg_vptr[s], tmp_e, tmp_e_bytes) && launch_col16_scan(st, entries, cost, cptr, tmp_e, tmp_e_bytes) &&
             hip_ok(hipMemcpyAsync(&p.seg_pieces[s], flag, sizeof(int), hipMemcpyDeviceToHost, st), "read piece flag") &&
             hip_ok(hipMemcpyAsync(&last[0], cptr + entries, sizeof(int), hipMemcpyDeviceToHost, st), "read pass cost") &&
             hip_ok(hipMemcpyAsync(&last[1], p.seg_vptr[s] + entries, sizeof(int), hipMemcpyDeviceToHost, st), "read pass size") &&
             hip_ok(hipStreamSynchronize(st), "sync run scans");
        total_cost = last[0];
      }
      if (ok && p.seg_pieces[s] > 0) { // the merge kernel's list: one wavefront per cut run (cnt_s / pos are still this slab's)
        int listed = 0;
        ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&p.seg_cut[s]), sizeof(int) * static_cast<size_t>(p.seg_pieces[s])), "hipMalloc cut runs") &&
             hip_ok(hipMemsetAsync(flag, 0, sizeof(int), st), "memset cut-run counter");
        if (ok) {
          launch_segment_cut_list(st, cnt_s, pos, A.m, kSegPiece, flag, p.seg_cut[s]);
          ok = hip_ok(hipMemcpyAsync(&listed, flag, sizeof(int), hipMemcpyDeviceToHost, st), "read cut-run count") &&
               hip_ok(hipStreamSynchronize(st), "sync cut runs");
          if (ok && listed != p.seg_pieces[s]) {
            set_error(kErrHip, "slab run lists: the cut-run list disagrees with the compaction");
            ok = false;
          }
        }
      }
      if (ok) {
        const int nblocks = segment_block_count(total_cost);
        ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&p.seg_blk[s]), sizeof(int) * (static_cast<size_t>(nblocks) + 1)), "hipMalloc pass workgroups");
        if (ok) {
          launch_segment_blocks(st, entries, nblocks, cptr, p.seg_blk[s]);
          ok = hip_ok(hipStreamSynchronize(st), "sync pass workgroups"); // (pieces / pos / cptr are reused or freed next)
          p.seg_blocks[s] = nblocks;
        }
      }
      for (void *q : {static_cast<void *>(len), static_cast<void *>(cost), static_cast<void *>(cptr), tmp_e})
        if (q) (void)hipFree(q);
      p.seg_entries[s] = entries;
      max_entries = entries > max_entries ? entries : max_entries;
      tune_log("m %d nnz %d: slab_segments: slab %d of %d: %d non-zeros in %d runs, %d workgroups", A.m, A.nnz, s, S, last[1], entries, p.seg_blocks[s]);
    }
    ok = ok && hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_seg_ys), sizeof(double) * (static_cast<size_t>(max_entries) + 1)), "hipMalloc run sums");
    if (ok && unordered == 2) {
      p.free_segments();
      p.seg_state = 0;
      tune_log("m %d nnz %d: slab_segments: too many short runs for 32-bit pass arithmetic, ordinary path", A.m, A.nnz);
    } else if (ok) {
      p.seg_state = 1;
      p.seg_slabs = S;
      p.seg_rest_below = rest_below;
    }
  }
  for (void *q : {static_cast<void *>(cnt), static_cast<void *>(beg), static_cast<void *>(pieces), static_cast<void *>(pos), static_cast<void *>(flag), tmp})
    if (q) (void)hipFree(q);
  if (!ok) {
    p.free_segments();
    return false;
  }
  return true;
}

} // namespace detail

} // namespace spmv_acc
