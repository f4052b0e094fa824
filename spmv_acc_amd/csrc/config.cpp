// config.cpp -- errors, the library stream, tunables, strategy names and the host-side pickers (split out of engine.cpp in round 4; no behaviour change).
//
// Reference roles: src/acc/strategy_picker.cpp:19-65 (dispatch), hip-adaptive/adaptive.cpp:16-67
// (adaptive decision), hip-flat/flat.cpp:30-57 (break-point staging), and
// hip-csr-adaptive-plus/csr_adaptive_plus_spmv.cpp:16-72 (analysis staging / destroy).
// The reference allocates (and for flat: leaks) its scratch on EVERY SpMV call and re-runs the
// preprocessing each time; here scratch and preprocessing results live in a plan that is created the
// first time a matrix (identified by its device pointers and shape) is seen, so the steady-state call
// is launches only: no hipMalloc / hipMemcpy / synchronisation, hipGraph-capturable.
#include "engine_internal.hpp"

namespace spmv_acc {
using namespace detail;

// ---- errors ----------------------------------------------------------------------------------------------
namespace detail {
thread_local int g_err = kOk;
thread_local std::string g_err_msg;
// The library stream belongs to the calling host thread, like HIP's current device: a process that drives N GPUs from N
// host threads (one hipSetDevice + one stream each; spmv-cli --gpus N, spmv_acc_sharded_spmv with ncclCommInitAll) gives every
// thread its own, and nothing one thread sets can redirect another thread's launches.  NULL (the reference's behaviour)
// until the thread calls spmv_acc_set_stream.
thread_local hipStream_t t_stream = nullptr;
std::mutex g_mu;

bool hip_ok(hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  set_error(kErrHip, std::string(what) + ": " + hipGetErrorString(e));
  return false;
}
} // namespace detail

void set_error(int code, const std::string &what) {
  g_err = code;
  g_err_msg = what;
}
namespace detail {
bool report_stale_last_plan(); // below, with the plan cache
}
// Besides the calling thread's own error this reports that the plan the calling thread used LAST is stale: a kernel of that call
// (or of an earlier one) found that the matrix behind the plan's pointers is no longer the one the plan was built for.  The
// kernels run asynchronously, so the flag is there only after the caller has synchronised -- as it must before reading y anyway:
// synchronise, then ask.  One relaxed load of a pinned int; no lock, no walk over the plan cache (other threads' plans are
// theirs to ask about; spmv_acc_check_plans() looks at every plan).  The stale plan is dropped; the next call rebuilds it.
int last_error() {
  if (g_err == kOk) (void)report_stale_last_plan();
  return g_err;
}
const char *last_error_string() { return g_err_msg.c_str(); }
void clear_error() {
  g_err = kOk;
  g_err_msg.clear();
}

namespace detail {
int last_error_code_only() { return g_err; }
thread_local double t_last_prepare_us = 0.0;
thread_local unsigned t_plan_work = 0; // bumped by every once-per-matrix step (structural pass, probe, timing) that really runs
} // namespace detail
unsigned plan_work_count() { return detail::t_plan_work; }
namespace detail {
} // namespace detail

double last_prepare_us() { return t_last_prepare_us; }

// The library stream is per host thread.  A consumer written for a process-wide stream (round 2's semantics) that sets it once on its main
// thread and calls from workers would launch on the NULL stream without any sign: the first such call says so, once per process, on stderr
// (SPMV_ACC_QUIET=1 silences it).  Threads that mean the NULL stream call spmv_acc_set_stream(NULL) themselves.
namespace detail {
thread_local bool t_stream_was_set = false;
std::atomic<bool> g_some_thread_set_a_stream{false};
void note_stream_use() {
  static std::atomic<bool> said{false};
  if (t_stream_was_set || !g_some_thread_set_a_stream.load(std::memory_order_relaxed) || said.exchange(true)) return;
  const char *quiet = std::getenv("SPMV_ACC_QUIET");
  if (quiet && *quiet && *quiet != '0') return;
  std::fprintf(stderr, "spmv_acc: a host thread that never called spmv_acc_set_stream is launching on the NULL stream while another thread has set a "
                       "stream -- the library stream is per host thread (include/spmv_acc.h); call spmv_acc_set_stream in every calling thread\n");
}
} // namespace detail
void set_stream(hipStream_t s) {
  t_stream = s;
  t_stream_was_set = true;
  if (s) g_some_thread_set_a_stream.store(true, std::memory_order_relaxed);
}
hipStream_t get_stream() { return t_stream; }

// ---- tunables (A/B switches for measurement; defaults are the shipped configuration) ----------------------
namespace detail {
#ifdef FLAT_SEGMENT_SUM_REDUCE
constexpr int kFlatReduceBuilt = 1;
#else
constexpr int kFlatReduceBuilt = 0;
#endif
Tunable g_tunables[] = {
    {"xcd_chunk", 16, 16},     // row blocks, flat tiles, row-block-plus blocks: each XCD takes this many consecutive blocks per super-chunk (0 = off; one knob
                               // since round 6: `xcd_chunk_tiles` is gone)
    {"rowblock_vec", 0, 0},    // 0 = pick from nnz/m, else force lanes per row
    {"rowblock_target", -1, -1}, // products a row block should bring to its 2048-product tile: -1 = 1800 and 1500 timed in turns once per plan, the faster stays (rule: 1800);
                               // a positive value pins it.  Round 1 measured 1900 best, round 3 1500 (1-2 %); with round 6's kernels 1800 is 3-6 % faster than
                               // 1500 on the stand-ins of 28 and more non-zeros per row and on the banded shard, 1500 1.5 % ahead at 12.6 per row, 1900
                               // 3-6 % behind (second rounds) -- profiles/r06_rowblock_target_sweep.txt
    {"stream_plain", -1, -1},  // stream-load cache policy: -1 = timed once per matrix; 0 nt, 1 default, 2 index default, 3 value default
    {"rowblock_guard", 1, 1},  // imbalance probe + rescue for the row-block family
    {"adaptive_timed", 1, 1},  // adaptive: 1 = time row blocks / row-block-plus / flat on the matrix and keep the fastest;
                               // 0 = decide from the four rowptr samples and the balance probe only
    {"adaptive_split", 0, 0},  // adaptive, halves differing >= 4x: 1 = the reference's two-width vector-row split
    {"plus_min_nnz", 0, 0},    // adaptive-plus analysis: MIN_NNZ_PER_BLOCK; 0 = time 1024 (the reference's instance) / 1536 /
                               // 1920 on the matrix and keep the fastest
    {"plus_host_analysis", 0, 0}, // 1: run the row-block analysis on the host (the reference's form)
    {"flat_finish", -1, -1},   // flat cut rows: -1 time both forms per matrix, 0 carries + fix-up kernel, 1 tiles finish them (when legal)
    {"flat_npt", -1, -1},      // non-zeros per lane per flat tile (tile = 256 lanes x this): 4, 8 or 16; -1 = 8, and below
                               // kFlatSmallNnz non-zeros 4 and 8 are both timed on the matrix (small grids: more, shorter workgroups)
    {"validate", 0, 0},        // 1: check rowptr / colindex of every new matrix on the device before the first launch
    {"rowlen", -1, -1},        // row-block family: row extents from the plan's 1-byte row lengths + per-block bases instead of
                               // rowptr: -1 where rows average <= 8 non-zeros (rowptr is then >= 3.5 % of the traffic), 0 never, 1 always
    {"flat_early", -1, -1},    // flat: issue a tile's stream loads before its break point -> rowptr chain: -1 timed per matrix
                               // below kFlatSmallNnz non-zeros (else off), 0 off, 1 on
    {"vector_tile", 1, 1},     // vector_row / light / the two-width split: 1 = w lanes per row over LDS-staged tiles (16-B stream
                               // loads), 0 = w lanes per row straight from global memory (4-/8-byte loads; also the form very
                               // uneven matrices keep)
    {"col16", -1, -1},         // 16-bit column encoding (k_col16.hip; row blocks and flat): -1 = built once per plan and timed against the caller's
                               // colindex per kernel family, kept where it wins by > 1.5 %; 0 never; 1 always where it can be built (16 / 32 / 64
                               // also pin the record size: tests).  A plan that uses it holds structure DERIVED from colindex: 64 samples of
                               // colindex are re-checked by every launch (like rowptr's), an in-place edit between the samples needs
                               // spmv_acc_release_plans -- `deterministic` and col16 = 0 keep to the caller's arrays.
    {"vector_width", 0, 0},    // vector_row / light: lanes per row; 0 = the reference's rule (vector_row.cpp:15-27: pow2 >= avg row length / 2)
    {"zigzag", 1, 1},          // every other SpMV on a plan walks the matrix in reverse block / tile order: with the streams cacheable, what the
                               // previous SpMV touched last is still in the 256 MB Infinity Cache when the next one starts there
                               // (Bump_2911-sized 156.5 -> 149 us, RM07R-sized 77.6 -> 74.2, largebasis-sized 15.5 -> 15.05; nothing where
                               // the plan streams non-temporally)
    {"cache_ends_mb", 24, 24}, // row blocks / flat under the non-temporal policy: MB of stream at each end of the grid that stay cacheable
                               // (an L2's worth: the zigzag order starts the next SpMV there); 0 = off
    {"flat_reduce", kFlatReduceBuilt, kFlatReduceBuilt}, // flat: how a tile's products become row sums: 0 = lane groups per row (w lanes per row from the tile's row
                               // count, long spans to whole waves), 1 = the segmented scan over the tile (the reference's option
                               // FLAT_SEGMENT_SUM_REDUCE, which as a build macro makes 1 the default like strategy_picker.cpp:34-39; what
                               // segment_sum_flat_sparse_spmv runs whatever this is set to); 2048-non-zero tiles.  Measured: +1 % on the
                               // headline matrix, +30..40 % on the small / medium stand-ins (three more barriers per tile)
    {"gather_hint", -1, -1},   // gather hints (k_hint.hip): the plan's column census marks the non-zeros whose x line is outside the hot set that
                               // fits an L2, and their gathers go non-temporal so they do not displace it.  -1 = where the census finds such a
                               // set (power-law columns) the kernel is timed with and without once per matrix; 0 off; 1 = always build and use
    {"hint_budget_kb", 4608, 4608}, // size of the hot set of x lines the census keeps cacheable.  An XCD's L2 is 4 MB; measured on R-MAT scale 25
                               // (8.17-8.21 ms without hints): 1 MB 9.43, 2 MB 8.33, 3 MB 7.71, 3.5 MB 7.35, 4 MB 7.06-7.25, 4.5 MB 7.21, 5 MB 7.19,
                               // 6 MB 7.25, 8 MB 7.70, 16 MB 7.94 ms -- a hot set smaller than what LRU keeps by itself loses, one around the L2 size wins
    {"deterministic", 0, 0},   // 1 (also: environment SPMV_ACC_DETERMINISTIC=1): NOTHING is timed.  Every choice the engine otherwise makes
                               // by timing on the matrix -- stream cache policy, adaptive's kernel family, flat's cut-row form / tile
                               // size / staging order, the row-block-plus block size, gather hints -- follows a fixed rule on the
                               // matrix' shape instead (strategy_picker.cpp:19-65: the reference's choice is a pure function of its
                               // inputs), so two processes run the same kernels in the same configuration and y is bitwise equal
                               // across processes and runs.  Costs the per-matrix optimum (a few per cent on most stand-ins).
                               // 0 (default, round 6): the rule serves UNTIL THE PLAN IS SETTLED -- the calls before that are answered by the plan's
                               // rule twin while the timings advance beside them against a scratch y, so early iterations are bitwise equal to each
                               // other; from the first settled call on the timed choices serve (dispatch.cpp run_spmv).  -1: the timed choices as
                               // far as they have come serve from the first call (rounds 2-5: the last bits of y could change while a plan settled)
    {"col_slabs", -1, -1},     // column-slab blocking with a slab-major COPY of colindex and values (k_slab.hip): A = sum of S column-range slabs, an SpMV is S
                               // consecutive SpMVs of the named strategy, each gathering from 1/S of x (power-law columns: R-MAT scale 25 5.0 -> 4.4 ms
                               // with S = 8; a slab keeps only the rows that have non-zeros in it).  -1 (round 6) = automatic: a plan whose own timed
                               // choice is the slab passes (slab_segments), once it has served 32 calls or inside spmv_acc_prepare, with free device
                               // memory >= 3 x 12 B per non-zero, builds the copy, times it against the passes and keeps the faster; the copy holds
                               // VALUES, so every call first compares 65,536 samples of the caller's values with the copy's (one small kernel and a
                               // stream synchronisation: such calls block) and refreshes the copy when they differ -- an in-place edit of fewer than
                               // ~1 in 10^4 values can slip through: call spmv_acc_refresh_values after such an edit, or set 0.  Never under
                               // `deterministic`, `strict_strategy` or inside a stream capture.  0 = never; S >= 2 = always, S slabs (then the caller
                               // refreshes: spmv_acc_refresh_values after editing values, spmv_acc_release_plans after editing colindex)
    {"flat_rowblock", -1, -1}, // flat on matrices whose fixed row blocks are balanced (nothing for non-zero-cut tiles to repair): -1 = time the flat tile
                               // kernel against the row-block kernel once per matrix and run the row blocks where they are >= 3 % faster (a flat tile
                               // needs one more dependent hop -- tile digest -> row extents -- and its cut rows a second kernel or a neighbour's carry:
                               // 7-11 % per launch on the small sweep stand-ins, 3-6 % on the large ones under the per-launch protocol); 0 = always
                               // the flat tile kernel; 1 = always the row blocks where balanced.  (Until late in round 3 only grids below 24 Mi
                               // non-zeros were timed.)  A caller that pins any of the tile kernel's own choices gets the tile kernel
    {"guard_full", 0, 0},      // OPT-IN: 1 = every SpMV re-reads ALL of rowptr and compares a 64-bit digest with the plan's (k_guard.hip) -- an
                               // in-place edit of the structure is then always noticed, not only where it touches one of the 64 samples of the
                               // guard the kernels carry.  4 * (m + 1) bytes and two small launches more per call
    {"slab_segments", -1, -1}, // column-slab blocking WITHOUT a copy of the matrix (k_segment.hip): where every row's columns ascend, the plan keeps
                               // per column slab the list of (row, first non-zero, length) runs -- structure only -- and an SpMV is S passes
                               // over those runs, each gathering from 1/S of x.  -1 = on matrices whose column census finds a hot set (the
                               // matrices that get gather hints: power-law columns, x far beyond the L2s) 8 slabs are built and timed once
                               // against the row-block-plus kernel, the faster stays (R-MAT scale 25: 7.3 -> 5.3 ms; with `deterministic`, which
                               // times nothing, the row-block-plus kernel stays); 0 = off; 1 = always, with the AUTOMATIC slab count (the x-size
                               // rule, tunable slab_kb: what tests use to reach the maximum count at test size); S >= 2 = always,
                               // whatever the strategy (rows that are not ordered: the ordinary path)
    {"first_call_budget", 20, 20}, // what the FIRST call on a matrix may spend on per-matrix timings, in SpMV-equivalents (wall time since the call began against
                               // N x the first trial launch it measured).  Once it is spent the call finishes by RULE -- every choice still open takes the
                               // `deterministic` rule for now and stays open -- and the following calls resume the timings, `later_call_budget`
                               // SpMV-equivalents each, until everything is settled.  0 = unbounded (rounds 1-3: 64 SpMVs' worth on the headline matrix).
                               // spmv_acc_prepare / spmv_acc_prepare_beta are always unbounded: they exist to pay for everything up front
    {"later_call_budget", 2, 2},   // see first_call_budget
    {"slab_whole_below", 32, 32}, // slab passes, two-class form (round 4): rows of fewer non-zeros than this are not cut by column slab at all -- each is ONE run,
                               // all columns, in a pass of its own after the S slab passes.  A row of d non-zeros gives ~min(d, 5.5) runs at S = 8; on R-MAT 25
                               // the rows below 32 non-zeros are 95 % of the rows, 12 % of the non-zeros and 35 M of the 45 M runs, and every run of one or two
                               // non-zeros is a 12-B list entry, a y read-modify-write and a part-used line of each stream: 5.30 -> 5.19 ms, first call 152 ->
                               // 135 ms (thresholds 8 / 16 / 24 / 32 / 48 / 64 / 128 / 256: 5.33 / 5.28 / 5.19 / 5.19 / 5.21 / 5.25 / 5.29 / 5.79,
                               // profiles/r04_rmat25_hub_windows_and_two_class.txt); 0 = every row is cut (round 3)
    {"strict_strategy", 0, 0}, // 1: a strategy NAME means its ALGORITHM, as in the reference (strategy_picker.cpp:19-65: the name IS the kernel): `flat` always runs
                               // flat_tile_kernel (no substitution of the row-block kernel on balanced rows, no row-block rescue of hypersparse tiles),
                               // `line_enhance` / `line` always the row-block kernel or, where fixed row blocks are unbalanced, its row-block-plus rescue
                               // (never the column-slab passes).  0 (default): the name selects a policy -- the engine may run whichever of its kernels it
                               // timed faster on the matrix.  adaptive / default are the engine's choice by definition and are not affected
    // ---- size rules, adjustable so that tests reach every size-selected branch at test size (round 5; defaults = the shipped rules) ----
    {"slab_kb", 32768, 32768}, // automatic slab count of the column-slab passes: KB of x per slab (S = x bytes / this, 2 ... 16: 16 from x = 496 MB on)
    {"hint_min_x_mb", 96, 96}, // the column census (gather hints, automatic slab passes) is taken from this many MB of x on (below, x lives in the Infinity Cache)
    {"max_grid_blocks", kMaxGridBlocks, kMaxGridBlocks}, // workgroups one launch may hold before a kernel whose grid grows with m strides over the rows
                               // (a launch holds < 2^32 work-items; 8,388,593 is prime).  Lowered by tests so that the striding runs at 10^5 rows
    {"flat_small_nnz_k", kFlatSmallNnz >> 10, kFlatSmallNnz >> 10}, // flat: below this many Ki non-zeros the tile size / staging order are timed per matrix (small grids)
};
static_assert(sizeof(g_tunables) / sizeof(g_tunables[0]) == kTunableCount, "TunableId must list every table entry, in order");
// (the count alone does not catch two entries in the wrong order -- round 4 ran an afternoon with first_call_budget reading slab_whole_below's
// value: the table's last names are checked against their ids once, at the first tunable lookup)
inline bool tunable_order_ok() {
  return std::strcmp(g_tunables[kT_first_call_budget].name, "first_call_budget") == 0 && std::strcmp(g_tunables[kT_later_call_budget].name, "later_call_budget") == 0 &&
         std::strcmp(g_tunables[kT_slab_whole_below].name, "slab_whole_below") == 0 && std::strcmp(g_tunables[kT_strict_strategy].name, "strict_strategy") == 0 &&
         std::strcmp(g_tunables[kT_flat_small_nnz_k].name, "flat_small_nnz_k") == 0 && std::strcmp(g_tunables[kT_max_grid_blocks].name, "max_grid_blocks") == 0 && std::strcmp(g_tunables[kT_hint_min_x_mb].name, "hint_min_x_mb") == 0 &&
         std::strcmp(g_tunables[kT_slab_segments].name, "slab_segments") == 0 && std::strcmp(g_tunables[kT_deterministic].name, "deterministic") == 0 &&
         std::strcmp(g_tunables[kT_zigzag].name, "zigzag") == 0 && std::strcmp(g_tunables[kT_xcd_chunk].name, "xcd_chunk") == 0 &&
         std::strcmp(g_tunables[kT_col16].name, "col16") == 0;
}
void apply_env_tunables();

} // namespace detail
// ---- kernel clock (kernels.hpp SPMV_ACC_LAUNCH): per host thread, a growing pool of event pairs, one pair per launch while the clock is on ----
namespace detail {
struct KernelClock {
  bool on = false;
  int device = -1;
  std::vector<hipEvent_t> pool; // pairs: [2k] start, [2k + 1] stop
  size_t used = 0;              // events handed out since kernel_clock_begin
  bool failed = false;
};
thread_local KernelClock t_kclock;
} // namespace detail
void kernel_clock_begin() {
  using detail::t_kclock;
  int dev = -1;
  (void)hipGetDevice(&dev);
  if (dev != t_kclock.device) { // (events belong to a device: a thread that moved on starts a new pool; the old events go with the process)
    t_kclock.pool.clear();
    t_kclock.device = dev;
  }
  t_kclock.used = 0;
  t_kclock.failed = false;
}
void kernel_clock_set(bool on) { detail::t_kclock.on = on; }
void kernel_clock_release() { // the timing entry is done with its events: the pool does not outlive the call
  for (hipEvent_t e : detail::t_kclock.pool) (void)hipEventDestroy(e);
  detail::t_kclock.pool.clear();
  detail::t_kclock.used = 0;
  detail::t_kclock.on = false;
}
size_t kernel_clock_used() { return detail::t_kclock.used; }
bool kernel_clock_failed() { return detail::t_kclock.failed; }
hipEvent_t kernel_clock_event(size_t i) { return i < detail::t_kclock.pool.size() ? detail::t_kclock.pool[i] : nullptr; }
bool kernel_clock_next(hipEvent_t *start, hipEvent_t *stop) {
  detail::KernelClock &k = detail::t_kclock;
  if (!k.on) return false;
  while (k.pool.size() < k.used + 2) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) {
      (void)hipGetLastError();
      k.failed = true;
      return false; // (the launch goes out plain; the timing entry reports the failure)
    }
    k.pool.push_back(e);
  }
  *start = k.pool[k.used];
  *stop = k.pool[k.used + 1];
  k.used += 2;
  return true;
}
int max_grid_blocks() {
  const int v = detail::g_tunables[detail::kT_max_grid_blocks].val;
  return v >= 64 && v <= kMaxGridBlocks ? v : kMaxGridBlocks;
}

namespace detail {
// SPMV_ACC_TUNABLES="validate=1,flat_finish=0": initial values for a process that cannot call spmv_acc_set_tunable
// (the reference's executables linked against this library).  Read once, before the first lookup.
void apply_env_tunables_once() {
  if (!tunable_order_ok()) {
    std::fprintf(stderr, "spmv_acc: internal error: tunable table and TunableId disagree\n");
    std::abort();
  }
  if (const char *det = std::getenv("SPMV_ACC_DETERMINISTIC"))
    if (*det && *det != '0') g_tunables[kT_deterministic].val = g_tunables[kT_deterministic].def = 1;
  const char *env = std::getenv("SPMV_ACC_TUNABLES");
  if (!env) return;
  std::string s(env);
  size_t pos = 0;
  while (pos < s.size()) {
    size_t end = s.find(',', pos);
    if (end == std::string::npos) end = s.size();
    const std::string item = s.substr(pos, end - pos);
    const size_t eq = item.find('=');
    if (eq != std::string::npos) {
      const std::string name = item.substr(0, eq);
      for (auto &t : g_tunables)
        if (name == t.name) t.val = t.def = std::atoi(item.c_str() + eq + 1);
    }
    pos = end + 1;
  }
}
void apply_env_tunables() {
  static std::once_flag once; // several host threads may make their first call together
  std::call_once(once, apply_env_tunables_once);
}
} // namespace detail

int set_tunable(const char *name, int value) {
  apply_env_tunables();
  for (auto &t : g_tunables) {
    if (std::strcmp(t.name, name) == 0) {
      // (the size-rule knobs that tests lower are clamped to what their arithmetic is written for: a negative MB count shifted left, a slab of
      // zero bytes, a threshold beyond int range -- ADVICE r05)
      const TunableId id = static_cast<TunableId>(&t - g_tunables);
      if (id == kT_hint_min_x_mb) value = value < 0 ? 0 : (value > (1 << 20) ? (1 << 20) : value);
      if (id == kT_slab_kb) value = value < 1 ? 1 : (value > (1 << 22) ? (1 << 22) : value);
      if (id == kT_flat_small_nnz_k) value = value < 1 ? 1 : (value > (1 << 21) ? (1 << 21) : value);
      if (id == kT_hint_budget_kb) value = value < 1 ? 1 : value;
      t.val = value;
      return 0;
    }
  }
  return -1;
}
int get_tunable(const char *name) {
  apply_env_tunables();
  for (auto &t : g_tunables)
    if (std::strcmp(t.name, name) == 0) return t.val;
  return -1;
}
void reset_tunables() {
  apply_env_tunables();
  for (auto &t : g_tunables) t.val = t.def;
}

// ---- strategy names -----------------------------------------------------------------------------------------
static const char *const kNames[kStrategyCount] = {"default",   "adaptive",     "thread_row", "wf_row",
                                                   "block_row_ordinary", "light", "vector_row", "line_enhance",
                                                   "line",      "flat",         "adaptive_plus"};

const char *strategy_name(int s) { return (s >= 0 && s < kStrategyCount) ? kNames[s] : "unknown"; }

int parse_strategy(const char *name) {
  if (!name) return -1;
  std::string s(name);
  for (auto &c : s) c = static_cast<char>(::tolower(static_cast<unsigned char>(c)));
  auto has = [&](const char *k) { return s.find(k) != std::string::npos; };
  // same test order as src/configure.cmake:18-37 ("line_enhance" before "line"); adaptive_plus is ours
  // and must be tested before "adaptive".
  if (has("adaptive_plus") || has("adaptive-plus")) return kAdaptivePlus;
  if (has("default")) return kDefault;
  if (has("adaptive")) return kAdaptive;
  if (has("thread_row")) return kThreadRow;
  if (has("wf_row")) return kWfRow;
  if (has("block_row_ordinary")) return kBlockRowOrdinary;
  if (has("light")) return kLight;
  if (has("vector_row")) return kVectorRow;
  if (has("line_enhance")) return kLineEnhance;
  if (has("line")) return kLine;
  if (has("flat")) return kFlat;
  return -1;
}

namespace detail {
int build_time_strategy() {
#if defined(KERNEL_STRATEGY_ADAPTIVE)
  return kAdaptive;
#elif defined(KERNEL_STRATEGY_THREAD_ROW)
  return kThreadRow;
#elif defined(KERNEL_STRATEGY_WAVEFRONT_ROW)
  return kWfRow;
#elif defined(KERNEL_STRATEGY_BLOCK_ROW_ORDINARY)
  return kBlockRowOrdinary;
#elif defined(KERNEL_STRATEGY_LIGHT)
  return kLight;
#elif defined(KERNEL_STRATEGY_VECTOR_ROW)
  return kVectorRow;
#elif defined(KERNEL_STRATEGY_LINE_ENHANCE)
  return kLineEnhance;
#elif defined(KERNEL_STRATEGY_LINE)
  return kLine;
#elif defined(KERNEL_STRATEGY_FLAT)
  return kFlat;
#elif defined(KERNEL_STRATEGY_DEFAULT)
  return kDefault;
#else
  return kAdaptive; // config.cmake:15 ships KERNEL_STRATEGY "DEFAULT"; the headline config is adaptive
#endif
}
int g_strategy = -1;
} // namespace detail

int active_strategy() {
  std::lock_guard<std::mutex> lk(g_mu);
  if (g_strategy < 0) {
    g_strategy = build_time_strategy();
    if (const char *e = std::getenv("SPMV_ACC_KERNEL_STRATEGY")) {
      const int s = parse_strategy(e);
      if (s >= 0) g_strategy = s;
    }
  }
  return g_strategy;
}

int set_active_strategy(int s) {
  if (s < 0 || s >= kStrategyCount) {
    set_error(kErrUnknownStrategy, "unknown strategy id");
    return -1;
  }
  std::lock_guard<std::mutex> lk(g_mu);
  g_strategy = s;
  return 0;
}

// ---- pickers --------------------------------------------------------------------------------------------------
int adaptive_branch(int m, const RowptrSamples &s) {
  const long long upper = s.half;           // nnz of rows [0, m/2)
  const long long lower = static_cast<long long>(s.last) - s.half; // nnz of rows [m/2, m)
  const long long big = upper > lower ? upper : lower;
  const long long small = upper > lower ? lower : upper;
  // "the two halves differ by 4x or more" (integer ratio as in adaptive.cpp:34-35; an empty half
  // counts as an unbounded ratio instead of dividing by zero)
  if (big != small && (small == 0 || big / small >= 4)) return 1;
  if (s.last / m <= 4) return 2;
  if (s.last <= 0xC00000) return 3;
  if (s.last > (1 << 23)) return 4;
  return 5;
}

namespace detail {
// do the four row quarters differ by 1.75x or more in non-zeros?
bool quarters_uneven(const RowptrSamples &s) {
  const long long q[4] = {s.q1, static_cast<long long>(s.half) - s.q1, static_cast<long long>(s.q3) - s.half,
                          static_cast<long long>(s.last) - s.q3};
  long long lo = q[0], hi = q[0];
  for (long long v : q) {
    lo = v < lo ? v : lo;
    hi = v > hi ? v : hi;
  }
  return lo <= 0 ? hi > 0 : 4 * hi >= 7 * lo;
}
// lanes per row by average row length: vector_row.cpp:15-27 / line_strategy.cpp:61-76
int classic_vec(long long avg) {
  if (avg <= 4) return 2;
  if (avg <= 8) return 4;
  if (avg <= 16) return 8;
  if (avg <= 32) return 16;
  if (avg <= 64) return 32;
  return 64;
}
} // namespace detail

namespace detail {
// lanes per row of the vector-row TILE kernel: same shape as classic_vec, 8 products per lane instead of 2 (the products are
// in LDS already; see k_vector_row.hip)
int tile_vec(long long avg) {
  int w = 2;
  while (w < 64 && 8LL * w < avg) w <<= 1;
  return w;
}
} // namespace detail

int plus_pick_vec(int m, int nnz) {
  const int avg = (m > 0) ? nnz / m : 0;
  int v = 1;
  while (v < 64 && avg > 2 * v) v <<= 1; // avg<=2 ->1, <=4 ->2, <=8 ->4, ... >64 ->64
  return v;
}

int plus_pick_vec_tuned(int m, int nnz, int min_nnz) {
  const long long avg = (m > 0) ? static_cast<long long>(nnz) / m : 0;
  // largest pow2 v with (THREADS / v) * avg >= 1.25 * MIN_NNZ (avg/5 for MIN_NNZ 1024): blocks then close on their
  // non-zero count, not on the row cap
  const long long per_vec = (5LL * min_nnz + 4 * kPlusThreads - 1) / (4 * kPlusThreads);
  int v = 1;
  while (v < 64 && static_cast<long long>(v) * 2 * per_vec <= avg) v <<= 1;
  return v;
}

// Host form of the row-block preprocessing pass.  Written as "where does the block that starts at
// row s end" so the device form can later replace the scan by searches; the emitted tables are
// bit-identical to the reference's single-pass loop (tests pin this against oracle/_ref).
int plus_analyze_host(int m, int min_nnz, int threads_per_block, int vec_size, const int *rp,
                      std::vector<int> &bp, std::vector<int> &fbr) {
  const int row_cap = threads_per_block / vec_size;
  const long long long_row = 2LL * min_nnz; // rows at least this long get dedicated blocks
  bp.clear();
  fbr.assign(static_cast<size_t>(m) + 1, 0);
  bp.push_back(0);
  int start = 0; // first row of the open block
  while (start < m) {
    // grow the open block one row at a time until it closes
    int row = start;
    long long acc = 0;
    for (;; ++row) {
      const long long len = static_cast<long long>(rp[row + 1]) - rp[row];
      acc += len;
      if (acc >= min_nnz) {
        if (len >= long_row) {
          const int slices = static_cast<int>(len / long_row);
          const bool alone = (acc == len); // nothing but this row's non-zeros in the open block
          for (int k = 0; k < slices; ++k) {
            if (!(k == 0 && alone)) bp.push_back(row);
            if (k == 0) fbr[row] = (static_cast<int>(bp.size()) - 1) * 2 + 1;
          }
        }
        bp.push_back(row + 1);
        break;
      }
      if (row - start + 1 >= row_cap || row == m - 1) {
        bp.push_back(row + 1);
        fbr[row + 1] = (static_cast<int>(bp.size()) - 1) * 2;
        break;
      }
    }
    start = row + 1;
  }
  return static_cast<int>(bp.size()) - 1;
}

} // namespace spmv_acc
