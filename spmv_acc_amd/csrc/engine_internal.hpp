// engine_internal.hpp -- what the four translation units of the engine share (config.cpp, plan.cpp, tuner.cpp, dispatch.cpp): the Plan, the
// tunables' ids, the per-call thread-local state, the timing helper and the two timing templates.  Not installed, not part of any API:
// everything here lives in spmv_acc::detail.  (Round 4: engine.cpp, 2,900 lines in one file, was split along these lines; no behaviour change.)
#pragma once

#include "engine.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <tuple>

#include <vector>

#include "kernels.hpp"

namespace spmv_acc {
namespace detail {

// ---- config.cpp: errors, stream, tunables ------------------------------------------------------------------------------------------------
extern thread_local hipStream_t t_stream;
void note_stream_use(); // one-time stderr note when a thread without a stream of its own launches while another thread has one
extern std::mutex g_mu; // the plan cache's lock (never held together with a plan's own)
bool hip_ok(hipError_t e, const char *what);
int last_error_code_only();
extern thread_local double t_last_prepare_us;
extern thread_local unsigned t_plan_work; // bumped by every once-per-matrix step (structural pass, probe, timing) that really runs
extern const char *const kStaleText;
struct Tunable {
  const char *name;
  int def;
  int val;
};
// indices into g_tunables, in table order (the kernels' hot path reads tunables by index, not by name)
enum TunableId {
  kT_xcd_chunk, kT_rowblock_vec, kT_rowblock_target, kT_stream_plain, 
  kT_rowblock_guard, kT_adaptive_timed, kT_adaptive_split, 
  kT_plus_min_nnz, kT_plus_host_analysis, kT_flat_finish, kT_flat_npt, kT_validate, kT_rowlen, kT_flat_early, kT_vector_tile, kT_col16, kT_vector_width, kT_zigzag, kT_cache_ends_mb, kT_flat_reduce, kT_gather_hint, kT_hint_budget_kb, kT_deterministic, kT_col_slabs, kT_flat_rowblock, kT_guard_full, kT_slab_segments, kT_first_call_budget, kT_later_call_budget, kT_slab_whole_below, kT_strict_strategy, kT_slab_kb, kT_hint_min_x_mb, kT_max_grid_blocks, kT_flat_small_nnz_k, kTunableCount
};
extern Tunable g_tunables[];
void apply_env_tunables();
// A plan's RULE TWIN (round 6, dispatch.cpp run_spmv): while a matrix's per-matrix timings are still open its calls are served by a second plan of the
// same arrays that decides everything by rule -- what `deterministic = 1` does for every call -- so that the iterations before the plan settles are
// bitwise equal to each other whatever the timings have found so far.  This thread is serving from (or building) such a twin:
extern thread_local bool t_rule_twin;
// tunable `deterministic`: 1 = every call by rule; 0 (default) = by rule until the plan is settled, the timed choices from then on; -1 = the timed
// choices as far as they have come serve from the first call (rounds 2-5).  tun(kT_deterministic) reads as a boolean everywhere: "is THIS call by rule?"
inline int tun(TunableId id) { // apply_env_tunables() has run: run_spmv calls it first
  if (id == kT_deterministic) return (t_rule_twin || g_tunables[id].val > 0) ? 1 : 0;
  return g_tunables[id].val;
}
inline bool rule_until_settled() { return g_tunables[kT_deterministic].val == 0; }
bool quarters_uneven(const RowptrSamples &s);
int classic_vec(long long avg);
int tile_vec(long long avg);

// ---- plan.cpp: the plan, its guard, the caches ---------------------------------------------------------------------------------------------
int guard_acquire(int device, const int **d_guard, int **h_flag);
void guard_release(int device, int slot, bool launched, hipStream_t last_stream);
enum Family { kFamRowblock = 0, kFamPlus = 1, kFamFlat = 2, kFamVector = 3, kFamilyCount = 4 };
// the kernel behind a plan's latest SpMV (Plan::last_kernel; include/spmv_acc.h SPMV_ACC_KERNEL_*)
enum LastKernel { kKernelRowblock = 0, kKernelPlus = 1, kKernelFlatTile = 2, kKernelSlabPasses = 3, kKernelVectorTile = 4, kKernelVectorRow = 5,
                  kKernelWaveRow = 6, kKernelLight = 7, kKernelBlockRow = 8, kKernelColSlabs = 9, kKernelScaleOnly = 10 };
extern thread_local int t_strict_name; // the strategy the caller NAMED when tunable strict_strategy is on (else -1): run_flat / run_plus keep to the name's kernel

typedef std::tuple<int, const void *, const void *, const void *, int, int, int> PlanKey; // device, rowptr, colindex, value, m, n, 1 for a rule twin

struct Plan {
  int device = 0;
  PlanKey key;                     // where the plan sits in g_plans
  hipStream_t last_stream = nullptr; // stream of the plan's latest launches (a plan may be used from several streams in turn)
  hipEvent_t order_event = nullptr;  // orders a call on another stream behind the plan's previous launches (run_spmv)
  bool launched = false;           // some kernel carrying this plan's guard slot has been enqueued
  unsigned long long last_use = 0; // plan-cache clock at the last call that used this plan
  std::mutex mu;                   // held by run_spmv for the whole call: plan fields, carry buffers and tunings are per matrix
  unsigned long long calls = 0;    // SpMV calls served by this plan (the first one builds and tunes it)
  unsigned launches = 0;           // tile-kernel launches so far (parity = walking direction, tunable zigzag)
  double trial_ms = 0.0;           // a trial launch of this matrix as the per-matrix timings measured it: prices later calls' tuning budget
  int last_c16 = -1;               // the plan's latest SpMV read the 16-bit column encoding (its record ints) or colindex (0): spmv_acc_query_plan_col16
  int last_kernel = -1;            // which kernel the plan's latest SpMV ran (kKernel*, below): spmv_acc_query_plan_last_kernel, strict_strategy's test
  bool tuning_open = true;         // some per-matrix timing was deferred (or has not been reached yet): later calls may resume it
  bool captured = false;           // a stream capture recorded kernels of this plan (a rule twin then outlives its plan's settling)
  unsigned settled_for[2] = {0, 0}; // bit s: a call of strategy s and this beta class has run on this plan with nothing left open on its path -- such calls time
                                   // nothing from then on, whatever other strategies / the other beta class still have open (run_spmv's rule-twin decision)
  CsrDev A;
  int guard_slot = -1;
  bool have_samples = false;
  RowptrSamples samples;
  // cache policy of the stream loads (kStreamPolicy*), timed once per matrix AND kernel family; -1 = not tuned yet
  // ... AND per beta class ([0]: beta == 0, y is only written; [1]: y is read as well): the extra 8 B/row change what the
  // streams should leave in the Infinity Cache -- on the Hardesty3-sized matrix without far columns the three policies tie at
  // beta = 0 (108 / 107 / 107 us) and differ by 7 % at beta = 1 (112 / 121 / 119 us)
  int stream_policy[kFamilyCount][2] = {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}};
  // opt-in structural check (tunable `validate`): -1 not run, 0 arrays are consistent, else the failure bits
  int invalid = -1;
  // row-block family: -1 unknown, 1 balanced, 0 some workgroup would need too many LDS rounds
  int rowblock_ok = -1;
  int rowblock_rpb = 0;
  int rb_target = 0; // timed choice between kRowblockTargetRule and kRowblockTargetAlt products per row block (0 = not timed: the rule)
  int max_block_nnz = 0;
  bool rowblock_uneven = false; // many row blocks far from the average block (balance probe)
  // adaptive's timed choice per beta class ([0]: beta == 0, [1]: y is read too -- the ranking flips between the classes where rows
  // hold one or two non-zeros): 0 fixed row blocks, 1 row-block-plus, 2 flat; -1 not timed yet
  int adaptive_family[2] = {-1, -1};
  float adaptive_ms[2][3] = {{1e30f, 1e30f, 1e30f}, {1e30f, 1e30f, 1e30f}}; // the comparison's timings per beta class (fixed row blocks, row-block-plus, flat)
  bool adaptive_provisional[2] = {false, false}; // the choice rests on the first look only (or on the families timed so far): later calls complete it
  bool adaptive_skipped[2][3] = {{false, false, false}, {false, false, false}}; // a family that is not a candidate on this matrix (rescued row blocks)
  RowDigest digest;             // row-block family: 1-byte row lengths + per-block bases (built for digest.rpb rows per block)
  // flat
  int flat_tiles = -1;
  FlatPlan flat;
  Col16 col16;                  // 16-bit column encoding (k_col16.hip), built the first time a family wants to time it (tunable col16: -1 timed, 0 never, 1 always)
  int c16_use[kFamilyCount] = {-1, -1, -1, -1}; // timed choice per kernel family: -1 not timed, 0 the caller's colindex, 1 the encoding (row blocks and flat only)
  int flat_npt_choice = 0;      // timed tile size (non-zeros per lane), 0 = not timed
  bool flat_geometry_tuned = false;
  int flat_rowblock_choice = -1;       // small grids: -1 not timed, 0 flat's own tile kernel, 1 the row-block kernel (tunable flat_rowblock)
  bool flat_early_choice = false;      // timed staging order (kept here as well: a FlatPlan is rebuilt when the tile size changes)
  int flat_mode_choice[2] = {-1, -1};  // timed cut-row form per beta class: -1 not timed, 0 tiles finish their cut rows, 1 carries + fix-up
  // persistent choices (tune cache): key of this matrix on this device, 0 = none
  unsigned long long tune_key = 0;
  // opt-in column-slab blocking (tunable col_slabs): the slabs' row pointers (S * (m + 1) ints), the re-ordered colindex / values,
  // where each slab starts in them, and each slab's non-zero count
  unsigned *d_light_counter = nullptr; // LIGHT's row counter (k_legacy.hip)
  // opt-in full row-pointer check (tunable guard_full, k_guard.hip): digest of rowptr[0 .. m] at plan-build time and the
  // arrays the per-call partial digests go to -- kDigestSlots of them, used in turn, so that calls on this plan that are in
  // flight on DIFFERENT streams at the same time do not share one
  static constexpr int kDigestSlots = 8;
  bool have_rp_digest = false;
  unsigned long long rp_digest = 0;
  unsigned long long *d_digest_acc = nullptr;
  unsigned digest_turn = 0;
  int slab_count = 0;
  int slab_width = 0;
  // the AUTOMATIC slab-major copy (round 6, dispatch.cpp::slab_copy_auto; tunable col_slabs = -1): -1 not looked at, 0 not used (no room, slower, refused), 1 in use;
  // the value samples that guard it (device) and the flag their comparison raises (pinned host memory)
  int slab_copy_choice = -1;
  unsigned long long *d_value_samples = nullptr;
  int *h_values_changed = nullptr;
  int value_samples = 0;
  unsigned values_refreshed = 0; // how often an in-place edit of the values was noticed and the copy refreshed
  long long *d_slab_off = nullptr; // the slabs' start positions, on the device (kept for spmv_acc_refresh_values)
  // per slab, COMPACT: the rows that have non-zeros in the slab (ascending ids), their row pointers (ms + 1), how many there are;
  // one scratch vector for a slab's compact result
  std::vector<int *> slab_rowid, slab_crp;
  std::vector<int> slab_rows;
  double *d_slab_ys = nullptr;
  int *d_slab_rp = nullptr;        // the slabs' DENSE row pointers (S * (m + 1)): where the scatter (and a values refresh) puts a non-zero
  int *d_slab_ci = nullptr;
  double *d_slab_v = nullptr;
  std::vector<long long> slab_off;
  // row-block-plus
  int plus_blocks = -1;
  int plus_vec = 0;
  int plus_min = 0; // MIN_NNZ_PER_BLOCK the analysis ran with
  int plus_tuned_min = 0; // the timed choice (0 = not timed yet)
  bool plus_has_long = false;
  // gather hints: census state (-1 not taken, 0 no hot set worth protecting / not applicable, 1 bits built), the bits, the timed
  // choice per kernel family (-1 not timed, 0 plain gathers, 1 hinted)
  int hint_state = -1;
  unsigned char *d_cold = nullptr;
  double hint_hot_share = 0.0;
  int hint_use[kFamilyCount] = {-1, -1, -1, -1};
  int *d_pbp = nullptr;
  int *d_pfbr = nullptr;
  double *d_ppartial = nullptr;
  void *d_pblk = nullptr;

  ~Plan() {
    free_device();
    guard_release(device, guard_slot, launched, last_stream);
  }
  bool is_stale() const { return A.stale && __atomic_load_n(A.stale, __ATOMIC_RELAXED) != 0; }
  void free_col16() {
    if (col16.d16) (void)hipFree(col16.d16);
    if (col16.rec) (void)hipFree(col16.rec);
    if (col16.ovf) (void)hipFree(col16.ovf);
    if (col16.ci_guard) (void)hipFree(col16.ci_guard);
    col16 = Col16();
  }
  void free_digest() {
    if (digest.lens) (void)hipFree(digest.lens);
    if (digest.base) (void)hipFree(digest.base);
    digest = RowDigest();
  }
  void free_slabs();
  // column-slab blocking without a copy (tunable slab_segments, k_segment.hip): -1 not looked at, 0 the rows are not slab-ordered
  // (ordinary path), 1 built for seg_slabs slabs
  int seg_state = -1, seg_slabs = 0;
  int seg_choice = -1; // automatic mode: -1 not timed, 0 the row-block-plus kernel stays, 1 the slab passes
  bool seg_early_tried = false; // the comparison against the COARSE row-block-plus kernel (dispatch.cpp::run_plus) has been made
  // per slab: one entry per run (or piece of a long run): its row, its first non-zero, its place in the pass's virtual non-zero
  // order (entries + 1 prefix sums of the lengths); and the first entry of every workgroup (blocks + 1)
  std::vector<int *> seg_row, seg_begin, seg_vptr, seg_blk, seg_cut; // seg_cut[s]: first entries of the slab's cut runs (seg_pieces[s] of them)
  std::vector<int> seg_entries, seg_blocks, seg_pieces; // seg_pieces[s]: runs of the slab that were cut into pieces (> 0: merge kernel needed)
  double *d_seg_ys = nullptr; // one partial sum per entry of the longest list
  int seg_rest_below = 0;      // two-class form: rows of fewer non-zeros than this are whole runs in the last plane (0: every row is cut by slab)
  int seg_whole_hint = -1;     // that plane's gathers with the plan's hints: -1 not timed (then: hinted), 0 plain, 1 hinted (dispatch.cpp::decide_whole_pass_hint)
  void free_segments() {
    seg_rest_below = 0;
    seg_whole_hint = -1;
    for (auto *list : {&seg_row, &seg_begin, &seg_vptr, &seg_blk, &seg_cut}) {
      for (int *q : *list)
        if (q) (void)hipFree(q);
      list->clear();
    }
    seg_entries.clear();
    seg_blocks.clear();
    seg_pieces.clear();
    if (d_seg_ys) (void)hipFree(d_seg_ys);
    d_seg_ys = nullptr;
    seg_state = -1;
    seg_slabs = 0;
  }
  void free_device() {
    if (order_event) (void)hipEventDestroy(order_event);
    order_event = nullptr;
    free_slabs();
    free_segments();
    if (d_light_counter) (void)hipFree(d_light_counter);
    d_light_counter = nullptr;
    if (d_digest_acc) (void)hipFree(d_digest_acc);
    d_digest_acc = nullptr;
    if (d_cold) (void)hipFree(d_cold);
    d_cold = nullptr;
    hint_state = -1;
    free_flat();
    free_digest();
    free_col16();
    if (d_pbp) (void)hipFree(d_pbp);
    if (d_pfbr) (void)hipFree(d_pfbr);
    if (d_ppartial) (void)hipFree(d_ppartial);
    if (d_pblk) (void)hipFree(d_pblk);
    d_pblk = nullptr;
    d_ppartial = nullptr;
    d_pbp = d_pfbr = nullptr;
  }
  static void free_flat_plan(FlatPlan &F) {
    if (F.bp) (void)hipFree(F.bp);
    if (F.head) (void)hipFree(F.head);
    if (F.tail) (void)hipFree(F.tail);
    if (F.tail_row) (void)hipFree(F.tail_row);
    if (F.tail_end) (void)hipFree(F.tail_end);
    if (F.digest) (void)hipFree(F.digest);
    F = FlatPlan();
  }
  void free_flat() {
    free_flat_plan(flat);
    flat_tiles = -1;
  }
};

struct TuneRecord {
  int v[26]; // stream_policy[4][2], adaptive_family[2], flat_npt, flat_early, flat_geometry_tuned, flat_mode[2], plus_min, hint_state0, hint_use[3], flat_rowblock, seg_choice, c16_use[row blocks], c16_use[flat], rb_target, slab_copy_choice
  bool operator==(const TuneRecord &o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
};
constexpr int kTuneFields = 26;
bool tune_cache_enabled();
void tune_adopt(Plan &p);
void tune_store(const Plan &p);
extern std::map<PlanKey, std::shared_ptr<Plan>> g_plans; // a running call keeps its plan alive through its own reference
extern thread_local std::weak_ptr<Plan> t_last_plan;     // the plan this thread's latest run_spmv used
extern thread_local bool t_capturing;
bool plan_work_allowed(const char *what);
// is there a plan of these arrays whose timings are closed for calls of this strategy and beta class? (run_spmv's rule-twin decision; no plan is made)
bool plan_settled_for(const int *rp, const int *ci, const double *v, int m, int n, int strategy, int cls);
void drop_rule_twin(const int *rp, const int *ci, const double *v, int m, int n);
bool rule_twin_exists(const int *rp, const int *ci, const double *v, int m, int n);
const int *host_view(const int *h); // h if the pointer is host-readable, else null
bool fetch_samples(Plan &p, const int *h_rowptr);
std::shared_ptr<Plan> get_plan(int m, int n, int nnz, const int *h_rowptr, const int *rp, const int *ci, const double *v);
bool report_stale_last_plan();
void drain_deferred_locked();

// ---- tuner.cpp: structural passes and per-matrix timings -----------------------------------------------------------------------------------
extern thread_local bool t_flat_segment_sum; // this thread is inside segment_sum_flat_sparse_spmv (FlatSegmentSumScope)
inline bool flat_segment_sum() { return (t_flat_segment_sum || tun(kT_flat_reduce) == 1) && tun(kT_col16) <= 0; }
extern thread_local int t_beta_class; // beta class of the call being served (set by run_spmv): [0] beta == 0, [1] y is read too
inline double trial_beta() { return t_beta_class ? 1.0 : 0.0; }
extern thread_local bool t_coarse_tuning;
extern thread_local bool t_no_policy_timing;
extern thread_local bool t_in_slab;
inline bool next_reverse(Plan &p) { return tun(kT_zigzag) && (p.launches++ & 1u); }
// products a row block should bring to its 2048-product tile: the tunable when it pins one, else the plan's timed choice, else the rule (dispatch.cpp::run_rowblock)
constexpr int kRowblockTargetRule = 1800, kRowblockTargetAlt = 1500;
inline int rowblock_target_for(const Plan &p) { return tun(kT_rowblock_target) > 0 ? tun(kT_rowblock_target) : (p.rb_target > 0 ? p.rb_target : kRowblockTargetRule); }
constexpr int kFlatSmallNnz = 24 << 20;
inline long long flat_small_nnz() { return static_cast<long long>(tun(kT_flat_small_nnz_k)) << 10; } // (tunable: tests cross the rule at test size)
int policy_for(const Plan &p, int fam);
bool tune_log_enabled();
void tune_log(const char *fmt, ...);
double *tune_scratch(size_t len);
void release_tune_scratch();
bool build_flat_plan(const CsrDev &A, int stride, hipStream_t stream, FlatPlan &F);
int flat_stride_for(const Plan &p);
bool ensure_flat(Plan &p, hipStream_t stream);
int analyze_on_device(hipStream_t st, const int *d_rowptr, int m, int min_nnz, int threads, int vec, int **d_bp, int **d_fbr);
bool ensure_plus(Plan &p, const int *h_rowptr, hipStream_t stream, int min_nnz);
bool ensure_col16(Plan &p, hipStream_t st);
bool ensure_hint(Plan &p, hipStream_t st);
bool ensure_slabs(Plan &p, int S, hipStream_t st);
bool ensure_segments(Plan &p, int S_cols, hipStream_t st);
void launch_flat_plan(hipStream_t st, const CsrDev &A, FlatPlan &F, int policy, double alpha, double beta, const double *x, double *y, bool reverse);
void launch_flat_with(hipStream_t st, Plan &p, int policy, double alpha, double beta, const double *x, double *y);
bool autotune_flat_mode(Plan &p, hipStream_t st, const double *x);
bool autotune_flat_geometry(Plan &p, hipStream_t st, const double *x);

// ---- plan-time budget (tunables first_call_budget / later_call_budget) ---------------------------------------------------------------
// The reference pays a fixed, small preprocessing cost per call (hip-flat/flat.cpp:39-44: one malloc + memset + break-point kernel); a plan
// that spends 64 SpMVs' worth of trial launches on its first call gives that advantage back to short solves.  So the trial launches of
// a call are bounded: run_spmv notes when the call began and how many SpMV-equivalents it may spend; the first trial launch measured in the
// call (TuneTimer) turns that into milliseconds; every timing PHASE asks defer_tuning() before it starts and, when the budget is
// spent, leaves its choice open (the `deterministic` rule serves the call) for a later call to settle.  Structural passes are not
// deferred -- a call cannot run without them -- but their time counts as spent.
extern thread_local std::chrono::steady_clock::time_point t_call_began;
extern thread_local double t_budget_spmvs; // SpMV-equivalents this call may spend; <= 0: unbounded
extern thread_local double t_budget_ms;     // the same in milliseconds, known once a trial launch has been measured in this call (or from the plan)
extern thread_local bool t_tuning_deferred; // some phase of this call left its choice open
extern thread_local float t_first_trial_ms;   // the first trial launch this call measured (0: none)
extern thread_local int t_unbounded_tuning;     // > 0: this thread is inside spmv_acc_prepare (UnboundedTuningScope): no budget
inline bool defer_tuning() {
  if (t_budget_spmvs <= 0.0 || t_budget_ms < 0.0) return false;
  const double spent = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call_began).count();
  // (a phase is atomic -- its candidates must be measured alike -- and costs 6-15 launches plus whatever structure it builds first, so phases only
  // START during the first half of the budget: the overshoot of the last one then lands near the whole)
  if (spent < 0.5 * t_budget_ms) return false;
  t_tuning_deferred = true;
  return true;
}
inline bool by_rule() { return tun(kT_deterministic) != 0 || defer_tuning(); }
inline double budget_spent_fraction() { // 0 while the call is unbounded or its budget has no price yet
  if (t_budget_spmvs <= 0.0 || t_budget_ms <= 0.0) return 0.0;
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call_began).count() / t_budget_ms;
}

struct TuneTimer {
  static constexpr int kMaxTimed = 5;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipEvent_t per[2 * kMaxTimed] = {};
  void *reset_ptr = nullptr;
  size_t reset_bytes = 0;
  bool ok = false;
  TuneTimer() {
    ok = hip_ok(hipEventCreate(&e0), "event") && hip_ok(hipEventCreate(&e1), "event");
    for (auto &e : per) ok = ok && hip_ok(hipEventCreate(&e), "event");
  }
  ~TuneTimer() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    for (auto &e : per)
      if (e) (void)hipEventDestroy(e);
  }
  // The trial launches write a scratch y.  With a reset buffer set the timed launches
  // follow the REFERENCE HARNESS'S protocol -- the one every figure of this repository is quoted on (benchmark/csr_spmv.hpp:66-74):
  // the scratch y is rewritten before each launch, each launch has its own event pair, the median counts -- instead of one event
  // pair around back-to-back launches.  Back-to-back timing favours whatever profits most from the previous launch's cache
  // contents and hides a second kernel's launch gap; candidates a few per cent apart ranked differently under the two protocols
  // (af_shell10-sized: adaptive kept fixed row blocks, 121.6 us per launch with y reset, where flat runs 118.2).
  void set_reset(void *ptr, size_t bytes) {
    reset_ptr = ptr;
    reset_bytes = bytes;
  }
  // at_least: launches a decision kept for the life of the plan rests on, whatever they cost (a launch of >= 4 ms is otherwise timed once)
  template <typename F> bool time(hipStream_t st, F &&fn, float *ms_per_launch, int at_least = 1) {
    if (!ok) return false;
    float first = 0.f;
    (void)hipEventRecord(e0, st);
    fn();
    (void)hipEventRecord(e1, st);
    if (!hip_ok(hipEventSynchronize(e1), "sync tune") || !hip_ok(hipEventElapsedTime(&first, e0, e1), "elapsed tune")) return false;
    if (t_first_trial_ms <= 0.f) t_first_trial_ms = first;
    if (t_budget_spmvs > 0.0 && t_budget_ms < 0.0) t_budget_ms = t_budget_spmvs * static_cast<double>(first); // (the call's first trial launch prices its budget)
    // (short kernels time noisily and cost nothing: more launches; from 0.1 ms on one more warm-up and three timed launches
    // separate candidates that differ by a few per cent -- the per-matrix timings of a 0.16 ms SpMV were 2/3 of a 21 ms first call)
    // (a launch of several milliseconds -- R-MAT scale 25: 7-8 ms, eighteen candidate launches = 145 ms of a 300 ms first call -- is its own
    // steady state: what the previous launch left in the caches is a fraction of a per cent of it.  One launch per candidate.)
    if (first >= 4.0f) {
      // (the first launch of a candidate runs right behind the build of its tables, the other candidate's state still in the caches:
      // a decision that is kept and persisted gets a second sample, the smaller counts)
      for (int extra = 1; extra < at_least; ++extra) {
        float again = 0.f;
        if (reset_ptr) (void)hipMemsetAsync(reset_ptr, 0, reset_bytes, st);
        (void)hipEventRecord(e0, st);
        fn();
        (void)hipEventRecord(e1, st);
        if (!hip_ok(hipEventSynchronize(e1), "sync tune") || !hip_ok(hipEventElapsedTime(&again, e0, e1), "elapsed tune")) return false;
        first = again < first ? again : first;
      }
      *ms_per_launch = first;
      return true;
    }
    // (under a call's tuning budget -- every call but spmv_acc_prepare's -- launches of >= 0.1 ms take no extra warm-up, the launch above was one, and
    // two timed launches instead of three: 3 launches per candidate instead of 5; the candidates of one phase are still measured alike)
    const bool lean = t_budget_spmvs > 0.0 && first >= 0.1f;
    const int warm = first < 0.1f ? 2 : (first < 2.0f && !lean ? 1 : 0);
    // (round 5: outside a budget -- spmv_acc_prepare -- five timed launches up to 0.5 ms and three up to 2 ms: the choices of a settled plan rest on
    // medians of five, where candidates 1-2 % apart used to change places between processes on medians of three)
    const int timed = first < 0.1f ? 5 : (first < 0.5f ? (lean ? 2 : 5) : (first < 2.0f ? (lean ? 2 : 3) : 1));
    for (int w = 0; w < warm; ++w) fn();
    if (reset_ptr) {
      for (int t = 0; t < timed; ++t) {
        (void)hipMemsetAsync(reset_ptr, 0, reset_bytes, st);
        (void)hipEventRecord(per[2 * t], st);
        fn();
        (void)hipEventRecord(per[2 * t + 1], st);
      }
      if (!hip_ok(hipEventSynchronize(per[2 * timed - 1]), "sync tune")) return false;
      float each[kMaxTimed];
      for (int t = 0; t < timed; ++t)
        if (!hip_ok(hipEventElapsedTime(&each[t], per[2 * t], per[2 * t + 1]), "elapsed tune")) return false;
      std::sort(each, each + timed);
      *ms_per_launch = each[timed == 2 ? 0 : timed / 2]; // median of 5 or 3; of two the smaller
      return true;
    }
    (void)hipEventRecord(e0, st);
    for (int t = 0; t < timed; ++t) fn();
    (void)hipEventRecord(e1, st);
    float ms = 0.f;
    if (!hip_ok(hipEventSynchronize(e1), "sync tune") || !hip_ok(hipEventElapsedTime(&ms, e0, e1), "elapsed tune")) return false;
    *ms_per_launch = ms / static_cast<float>(timed);
    return true;
  }
  // Ranking candidates that are a few per cent apart (round 5): `rounds` rounds with the candidates TAKING TURNS inside each round -- two blocks of
  // samples taken one after the other rank by the moment they were taken (clocks, what the other candidate left in the caches) as much as by the
  // kernel --, each turn one time() (a median of several launches under the harness protocol); per candidate the median over its rounds counts.
  // launch(c) launches candidate c; skip[c] (may be null) leaves a candidate out.  ms[c] is written for the candidates that ran.
  template <typename F> bool time_in_turns(hipStream_t st, int n, F &&launch, int rounds, float *ms, const bool *skip = nullptr) {
    constexpr int kMaxRounds = 5;
    constexpr int kMaxCandidates = 4;
    if (n > kMaxCandidates) return false;
    rounds = rounds < 1 ? 1 : (rounds > kMaxRounds ? kMaxRounds : rounds);
    float sample[kMaxCandidates][kMaxRounds];
    for (int r = 0; r < rounds; ++r)
      for (int c = 0; c < n; ++c) {
        if (skip && skip[c]) continue;
        if (!time(st, [&] { launch(c); }, &sample[c][r])) return false;
      }
    for (int c = 0; c < n; ++c) {
      if (skip && skip[c]) continue;
      std::sort(sample[c], sample[c] + rounds);
      ms[c] = rounds == 2 ? sample[c][0] : sample[c][rounds / 2];
    }
    return true;
  }
};
// rounds of a ranking between near-equal candidates: three where the call may spend (spmv_acc_prepare), one under a call's tuning budget
inline int ranking_rounds() { return t_budget_spmvs > 0.0 ? 1 : 3; }

// Time the stream-load cache policies on THIS matrix with the kernel family that will run it (scratch y, beta = 0:
// no side effects on the caller's y) and keep the fastest.  Up to eight launches per candidate (TuneTimer: 3 to reach
// that policy's cache steady state + 5 timed; 2 in all when a launch takes milliseconds), once per matrix.
template <typename Launch> bool autotune_policy(Plan &p, int fam, hipStream_t st, Launch &&launch) {
  const int cls = t_beta_class;
  if (p.stream_policy[fam][cls] >= 0) return true;
  if (tun(kT_stream_plain) >= 0 || by_rule() || t_no_policy_timing) return true; // pinned (A/B runs) / by rule (or the call's tuning budget is spent): policy_for decides, nothing is recorded
  // A matrix prepared in one beta class (spmv_acc_prepare: beta = 1) and then CAPTURED into a hipGraph in the other: timing would
  // synchronise inside the capture.  The call runs under the policy the other class measured (policy_for's fallback) and this
  // class is timed by the first call made outside a capture.
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &capture) == hipSuccess && capture != hipStreamCaptureStatusNone) return true;
  (void)hipGetLastError();
  // While adaptive compares the families only the first one times the three policies; the others run under that result
  // (policy_for) and the family that wins times its own on its next call.  (Timing all three per family made adaptive's
  // first call 21 ms on the Hardesty3-sized matrix, 134 SpMVs' worth; the comparison itself needs 8 launches per family.)
  if (t_coarse_tuning) {
    for (int f = 0; f < kFamilyCount; ++f)
      if (p.stream_policy[f][cls] >= 0) return true;
  }
  ++t_plan_work;
  double *scratch = nullptr;
  if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
  TuneTimer timer;
  timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
  // (zeroed: in the beta != 0 class the trial launches accumulate into it)
  bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
  const int candidates[3] = {kStreamPolicyNt, kStreamPolicyDefault, kStreamPolicyValueDefault};
  float best = 1e30f;
  int best_policy = kStreamPolicyNt;
  for (int c = 0; ok && c < 3; ++c) {
    // the policies differ through what they leave in the Infinity Cache for the NEXT SpMV, so each candidate first
    // runs until the caches hold its own steady state, then is timed over several launches
    float ms = 0.f;
    ok = timer.time(st, [&] { launch(candidates[c], scratch); }, &ms);
    if (ok) tune_log("m %d nnz %d family %d beta class %d: stream policy %d -> %.2f us", p.A.m, p.A.nnz, fam, cls, candidates[c], ms * 1e3f);
    if (ok && ms < best) {
      best = ms;
      best_policy = candidates[c];
    }
  }
  if (ok) tune_log("m %d nnz %d family %d: keeps stream policy %d", p.A.m, p.A.nnz, fam, best_policy);
  if (ok) p.stream_policy[fam][cls] = best_policy;
  return ok;
}

// Hints for kernel family `fam`: forced by the tunable, else timed once per matrix (with / without) and kept if they win by > 2 %.
// `launch(ys)` launches the family's kernel with the plan's current settings writing to ys; p.A.cold selects the hinted variant.
template <class Launch> bool autotune_hint(Plan &p, int fam, hipStream_t st, Launch launch) {
  p.A.cold = nullptr;
  const int mode = tun(kT_gather_hint);
  if (mode == 0) return true;
  if (p.hint_state < 0 || (mode < 0 && p.hint_state == 1 && p.hint_use[fam] < 0)) { // census / timing ahead: not inside a capture
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return true;
  }
  if (!ensure_hint(p, st)) return false;
  if (p.hint_state != 1) return true;
  if (mode > 0 || tun(kT_deterministic)) { // forced / by rule: wherever the census found a hot set worth protecting
    p.A.cold = p.d_cold;
    return true;
  }
  if (p.hint_use[fam] < 0 && defer_tuning()) return true; // (plain gathers for now; timed by a later call)
  if (p.hint_use[fam] < 0) {
    ++t_plan_work;
    double *scratch = nullptr;
    if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
    TuneTimer timer;
    timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
    bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
    float ms[2] = {0.f, 0.f};
    for (int h = 0; ok && h < 2; ++h) {
      p.A.cold = h ? p.d_cold : nullptr;
      ok = timer.time(st, [&] { launch(scratch); }, &ms[h]);
    }
    p.A.cold = nullptr;
    if (!ok) return false;
    p.hint_use[fam] = ms[1] < 0.98f * ms[0] ? 1 : 0;
    tune_log("m %d nnz %d family %d beta class %d gather hints: plain %.2f us, hinted %.2f us -> %s (kept for both classes)", p.A.m, p.A.nnz, fam,
             t_beta_class, ms[0] * 1e3f, ms[1] * 1e3f, p.hint_use[fam] ? "hinted" : "plain");
  }
  p.A.cold = p.hint_use[fam] == 1 ? p.d_cold : nullptr;
  return true;
}

// The 16-bit column encoding for kernel family `fam` (row blocks, flat): forced by the tunable, else built once per plan and timed once per family
// against the caller's colindex -- in turns, like every ranking of near-equal candidates -- and kept where it wins by > 1.5 %.
// `launch(c16, ys)` launches the family's kernel with the plan's current settings writing to ys, c16 = the encoding or null.  *out: what this
// call's launch should use.  Not with gather hints (p.A.cold set: power-law columns, nothing local to encode), not by rule (`deterministic`:
// the caller's arrays as they are), not inside a capture unless everything is decided, not while adaptive is still comparing families.
template <class Launch> bool autotune_col16(Plan &p, int fam, hipStream_t st, Launch launch, const Col16 **out) {
  *out = nullptr;
  const int mode = tun(kT_col16);
  if (mode == 0 || p.A.cold != nullptr || p.col16.state == 0) return true;
  if (mode < 0 && (tun(kT_deterministic) || p.c16_use[fam] == 0)) return true;
  const bool undecided = p.col16.state < 0 || (mode < 0 && p.c16_use[fam] < 0);
  if (undecided) {
    if (t_capturing || (mode < 0 && (t_coarse_tuning || t_no_policy_timing || defer_tuning()))) return true;
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone) return true;
    (void)hipGetLastError();
  }
  if (!ensure_col16(p, st)) return false;
  if (p.col16.state != 1) return true;
  if (mode < 0 && p.c16_use[fam] < 0) {
    ++t_plan_work;
    double *scratch = nullptr;
    if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
    TuneTimer timer;
    timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
    bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
    float ms[2] = {0.f, 0.f};
    ok = ok && timer.time_in_turns(st, 2, [&](int c) { launch(c ? &p.col16 : nullptr, scratch); }, ranking_rounds(), ms);
    if (!ok) return false;
    p.c16_use[fam] = ms[1] < 0.985f * ms[0] ? 1 : 0;
    tune_log("m %d nnz %d family %d beta class %d 16-bit columns (%d ints per chunk record, %lld escapes, %lld in overflow): colindex %.2f us, encoding %.2f us -> %s",
             p.A.m, p.A.nnz, fam, t_beta_class, p.col16.rec_ints, p.col16.escapes, p.col16.overflow, ms[0] * 1e3f, ms[1] * 1e3f,
             p.c16_use[fam] ? "encoding" : "colindex");
  }
  if (mode > 0 || p.c16_use[fam] == 1) *out = &p.col16;
  return true;
}

// ---- dispatch.cpp: the strategy runners ----------------------------------------------------------------------------------------------------
bool validate_plan(Plan &p, hipStream_t st);
bool launch_full_guard(Plan &p, hipStream_t st);
bool probe_rowblock(Plan &p, int rpb, hipStream_t st);
bool ensure_digest(Plan &p, int rpb, hipStream_t st);
bool run_flat(hipStream_t st, Plan &p, double alpha, double beta, const double *x, double *y);
bool run_rowblock(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y,
                  bool allow_uneven_switch = false, int lanes_per_row = 0);
bool run_plus(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y);
void run_segments(hipStream_t st, Plan &p, double alpha, double beta, const double *x, double *y);
int seg_auto_slabs(int n);

} // namespace detail
} // namespace spmv_acc
