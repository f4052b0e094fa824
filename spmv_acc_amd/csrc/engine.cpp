// engine.cpp -- plan cache, strategy dispatch and the host form of the row-block preprocessing pass.
//
// Reference roles: src/acc/strategy_picker.cpp:19-65 (dispatch), hip-adaptive/adaptive.cpp:16-67
// (adaptive decision), hip-flat/flat.cpp:30-57 (break-point staging), and
// hip-csr-adaptive-plus/csr_adaptive_plus_spmv.cpp:16-72 (analysis staging / destroy).
// The reference allocates (and for flat: leaks) its scratch on EVERY SpMV call and re-runs the
// preprocessing each time; here scratch and preprocessing results live in a plan that is created the
// first time a matrix (identified by its device pointers and shape) is seen, so the steady-state call
// is launches only: no hipMalloc / hipMemcpy / synchronisation, hipGraph-capturable.
#include "engine.hpp"

#include <hip/hip_runtime.h>

#include <algorithm>
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

namespace spmv_acc {

// ---- errors ----------------------------------------------------------------------------------------------
namespace {
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
} // namespace

void set_error(int code, const std::string &what) {
  g_err = code;
  g_err_msg = what;
}
namespace {
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

namespace {
int last_error_code_only() { return g_err; }
thread_local double t_last_prepare_us = 0.0;
thread_local unsigned t_plan_work = 0; // bumped by every once-per-matrix step (structural pass, probe, timing) that really runs
} // namespace

double last_prepare_us() { return t_last_prepare_us; }

void set_stream(hipStream_t s) { t_stream = s; }
hipStream_t get_stream() { return t_stream; }

// ---- tunables (A/B switches for measurement; defaults are the shipped configuration) ----------------------
namespace {
struct Tunable {
  const char *name;
  int def;
  int val;
};
// indices into g_tunables, in table order (the kernels' hot path reads tunables by index, not by name)
enum TunableId {
  kT_xcd_remap, kT_xcd_chunk, kT_xcd_chunk_tiles, kT_rowblock_vec, kT_rowblock_target, kT_stream_plain, kT_copy_nt,
  kT_stage_fast, kT_early_y, kT_rowblock_guard, kT_adaptive_timed, kT_adaptive_split, kT_rescue_flat, kT_plus_ref_vec,
  kT_plus_min_nnz, kT_plus_host_analysis, kT_flat_finish, kT_flat_npt, kT_validate, kT_rowlen, kT_flat_early, kT_vector_tile, kT_col16, kT_vector_width, kT_zigzag, kT_cache_ends_mb, kT_flat_reduce, kT_gather_hint, kT_hint_budget_kb, kT_deterministic, kT_tune_protocol, kT_col_slabs, kT_flat_rowblock, kT_legacy_kernels, kT_guard_full, kT_slab_segments, kT_vector_target, kT_first_call_budget, kT_later_call_budget, kT_slab_whole_below, kTunableCount
};
#ifdef FLAT_SEGMENT_SUM_REDUCE
constexpr int kFlatReduceBuilt = 1;
#else
constexpr int kFlatReduceBuilt = 0;
#endif
Tunable g_tunables[] = {
    {"xcd_remap", 0, 0},       // row-block family: XCD-contiguous block order (A/B: -1% .. +4% time; off)
    {"xcd_chunk", 16, 16},     // row-block family: each XCD takes this many consecutive blocks per super-chunk (0 = off)
    {"xcd_chunk_tiles", 16, 16}, // same order for the flat / row-block-plus grids (A/B after the cache-policy autotune,
                               // 9 stand-ins: 0 .. -4 % time on every one, none slower)
    {"rowblock_vec", 0, 0},    // 0 = pick from nnz/m, else force lanes per row
    {"rowblock_target", 1500, 1500}, // products a row block should bring to its 2048-product tile.  Round 1 measured 1900 best (fullest tile); with
                               // round 2-3's kernels (zigzag, per-matrix cache policy) and the per-launch protocol 1500 is 1-2 % faster on ten of
                               // eleven sweep stand-ins and 2.8 % on the banded shard, +0.8 % on TSOPF (tools/param_sweep_reset.py, fresh plans on
                               // the same arrays, profiles/r03_rowblock_target.txt); 1600 / 1400 / 1300 / 1700 / 2040 are not better
    {"stream_plain", -1, -1},  // stream-load cache policy: -1 = timed once per matrix; 0 nt, 1 default, 2 index default, 3 value default
    {"copy_nt", 1, 1},         // copy-ceiling probe: non-temporal loads/stores (0 = default cache policy)
    {"stage_fast", 1, 1},      // tile staging: wave-skip + branch-free form (0: per-lane predicated loads)
    {"early_y", 1, 1},         // row-block kernel: load the old y before the tile instead of after it
    {"rowblock_guard", 1, 1},  // imbalance probe + rescue for the row-block family
    {"adaptive_timed", 1, 1},  // adaptive: 1 = time row blocks / row-block-plus / flat on the matrix and keep the fastest;
                               // 0 = decide from the four rowptr samples and the balance probe only
    {"adaptive_split", 0, 0},  // adaptive, halves differing >= 4x: 1 = the reference's two-width vector-row split
    {"rescue_flat", 0, 0},     // 1: the rescue is flat (nnz-cut tiles) instead of the row-block-plus kernel
    {"plus_ref_vec", 0, 0},    // 1: row-block-plus analysis with the reference's VEC_SIZE pick (pow2 >= avg/2)
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
    {"col16", 0, 0},           // OPT-IN, flat only: 1 = the plan holds a 16-bit encoding of colindex (per-256-non-zero base + escape list,
                               // k_col16.hip) and the tile kernel streams 2 B instead of 4 B per column.  The plan then holds a copy
                               // derived from colindex: after editing colindex in place call spmv_acc_release_plans.
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
                               // across processes and runs.  Costs the per-matrix optimum (a few per cent on most stand-ins)
    {"tune_protocol", 1, 1},   // how the per-matrix timings are taken: 1 = the reference harness's protocol (y rewritten before each launch, one
                               // event pair per launch, median), 0 = one event pair around back-to-back launches (rounds 1-2)
    {"col_slabs", 0, 0},       // OPT-IN column-slab blocking (k_slab.hip): S >= 2 = the plan holds a re-ordered COPY of colindex and values,
                               // A = sum of S column-range slabs, and an SpMV is S consecutive SpMVs of the named strategy, each gathering
                               // from 1/S of x (power-law columns: the L2s then hold a hot set S times deeper; R-MAT scale 25 7.2 -> 4.4 ms
                               // with S = 8; a slab keeps only the rows that have non-zeros in it).  Costs S passes over y; loses on matrices whose gathers already hit.  After editing values
                               // in place call spmv_acc_refresh_values, after editing colindex spmv_acc_release_plans.  0 = off (the
                               // default: plans hold no copy of the matrix)
    {"flat_rowblock", -1, -1}, // flat on matrices whose fixed row blocks are balanced (nothing for non-zero-cut tiles to repair): -1 = time the flat tile
                               // kernel against the row-block kernel once per matrix and run the row blocks where they are >= 3 % faster (a flat tile
                               // needs one more dependent hop -- tile digest -> row extents -- and its cut rows a second kernel or a neighbour's carry:
                               // 7-11 % per launch on the small sweep stand-ins, 3-6 % on the large ones under the per-launch protocol); 0 = always
                               // the flat tile kernel; 1 = always the row blocks where balanced.  (Until late in round 3 only grids below 24 Mi
                               // non-zeros were timed.)  A caller that pins any of the tile kernel's own choices gets the tile kernel
    {"legacy_kernels", 1, 1},  // KERNEL_STRATEGY LIGHT / BLOCK_ROW_ORDINARY / THREAD_ROW: 1 = what the names mean in the reference (k_legacy.hip: rows
                               // handed out by an atomic counter; one workgroup per row; THREAD_ROW: one lane per row at every row length), 0 = the
                               // round-1/2 stand-ins (the vector-row tile kernel; one wavefront per row; the row-block kernel's own lanes-per-row
                               // pick), which are faster on most matrices (THREAD_ROW: by 0-4.5 %)
    {"guard_full", 0, 0},      // OPT-IN: 1 = every SpMV re-reads ALL of rowptr and compares a 64-bit digest with the plan's (k_guard.hip) -- an
                               // in-place edit of the structure is then always noticed, not only where it touches one of the 64 samples of the
                               // guard the kernels carry.  4 * (m + 1) bytes and two small launches more per call
    {"slab_segments", -1, -1}, // column-slab blocking WITHOUT a copy of the matrix (k_segment.hip): where every row's columns ascend, the plan keeps
                               // per column slab the list of (row, first non-zero, length) runs -- structure only -- and an SpMV is S passes
                               // over those runs, each gathering from 1/S of x.  -1 = on matrices whose column census finds a hot set (the
                               // matrices that get gather hints: power-law columns, x far beyond the L2s) 8 slabs are built and timed once
                               // against the row-block-plus kernel, the faster stays (R-MAT scale 25: 7.3 -> 5.3 ms; with `deterministic`, which
                               // times nothing, the row-block-plus kernel stays); 0 = off; S >= 2 = always,
                               // whatever the strategy (rows that are not ordered: the ordinary path)
    {"vector_target", 1900, 1900}, // vector-row tile kernel: products a workgroup's rows should bring to its 2048-product tile (the row-block family's
                               // `rowblock_target` went to 1500 in round 3; this kernel, with two rows per lane group, keeps the fuller tile:
                               // 1900 against 1500 is 3-5.5 % faster on four of five sweep stand-ins, equal on the fifth)
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
};
static_assert(sizeof(g_tunables) / sizeof(g_tunables[0]) == kTunableCount, "TunableId must list every table entry, in order");
// (the count alone does not catch two entries in the wrong order -- round 4 ran an afternoon with first_call_budget reading slab_whole_below's
// value: the table's last names are checked against their ids once, at the first tunable lookup)
inline bool tunable_order_ok() {
  return std::strcmp(g_tunables[kT_first_call_budget].name, "first_call_budget") == 0 && std::strcmp(g_tunables[kT_later_call_budget].name, "later_call_budget") == 0 &&
         std::strcmp(g_tunables[kT_slab_whole_below].name, "slab_whole_below") == 0 && std::strcmp(g_tunables[kT_vector_target].name, "vector_target") == 0 &&
         std::strcmp(g_tunables[kT_slab_segments].name, "slab_segments") == 0 && std::strcmp(g_tunables[kT_deterministic].name, "deterministic") == 0 &&
         std::strcmp(g_tunables[kT_zigzag].name, "zigzag") == 0 && std::strcmp(g_tunables[kT_xcd_remap].name, "xcd_remap") == 0;
}
void apply_env_tunables();
inline int tun(TunableId id) { return g_tunables[id].val; } // apply_env_tunables() has run: run_spmv calls it first

} // namespace

namespace {
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
} // namespace

int set_tunable(const char *name, int value) {
  apply_env_tunables();
  for (auto &t : g_tunables) {
    if (std::strcmp(t.name, name) == 0) {
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

namespace {
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
} // namespace

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

namespace {
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
} // namespace

namespace {
// lanes per row of the vector-row TILE kernel: same shape as classic_vec, 8 products per lane instead of 2 (the products are
// in LDS already; see k_vector_row.hip)
int tile_vec(long long avg) {
  int w = 2;
  while (w < 64 && 8LL * w < avg) w <<= 1;
  return w;
}
} // namespace

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

// ---- plans ---------------------------------------------------------------------------------------------------------
namespace {

// ---- stale-plan guard -------------------------------------------------------------------------------------------
// The reference recomputes its preprocessing on every call (flat.cpp:39-44), so it can never act on a matrix that has
// changed; its callers therefore never announce a change.  Plans here are keyed by pointers and shape, and a caller that
// frees a matrix and gets the same addresses back for another one of the same shape (the norm for hipMalloc after hipFree),
// or rewrites the structure in place, would meet the old matrix' break points / row blocks.  Every plan therefore records
// kGuardSamples rowptr entries (device) and owns a sticky flag in pinned host memory; the first wave of block 0 of every
// SpMV kernel compares (device_utils.hpp::check_plan_guard) and raises the flag.  The host looks at the flag -- an
// ordinary memory read, no synchronisation -- when the plan is used again and in spmv_acc_last_error(): the plan is
// dropped, SPMV_ACC_ERR_BAD_ARGUMENT is recorded (the y of the call that raised the flag is not to be trusted) and the
// matrix gets a fresh plan.  Slots come from one pool per device, recycled first-in first-out so that a kernel of a
// dropped plan that is still in flight does not meet its slot's next owner.
constexpr int kGuardSlots = 4096;
struct GuardPool {
  int *d_guard = nullptr; // kGuardSlots * kGuardSamples ints
  int *h_flags = nullptr; // kGuardSlots ints, hipHostMalloc (coherent, device-visible)
  // free slots, oldest first.  A slot released by a plan that had launched kernels carries an event recorded behind the plan's
  // last launch: the slot gets a new owner only once that event has completed, so a kernel of the dropped plan that is still
  // in flight can never raise the flag of the slot's next owner (first-in first-out alone only made that unlikely).
  std::deque<std::pair<int, hipEvent_t>> free_slots;
  bool failed = false;
};
std::mutex g_guard_mu; // not g_mu: plans die (and return their slot) both under g_mu and outside it
std::map<int, GuardPool> g_guard_pools;

int guard_acquire(int device, const int **d_guard, int **h_flag) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  GuardPool &P = g_guard_pools[device];
  if (P.failed) return -1;
  if (!P.d_guard) {
    if (hipMalloc(reinterpret_cast<void **>(&P.d_guard), sizeof(int) * kGuardSlots * kGuardSamples) != hipSuccess ||
        hipHostMalloc(reinterpret_cast<void **>(&P.h_flags), sizeof(int) * kGuardSlots,
                      hipHostMallocCoherent | hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) {
      (void)hipGetLastError();
      if (P.d_guard) (void)hipFree(P.d_guard);
      P.d_guard = nullptr;
      P.failed = true; // plans of this device run unguarded
      return -1;
    }
    std::memset(P.h_flags, 0, sizeof(int) * kGuardSlots);
    for (int i = 0; i < kGuardSlots; ++i) P.free_slots.emplace_back(i, nullptr);
  }
  int slot = -1;
  for (size_t tries = P.free_slots.size(); tries > 0 && slot < 0; --tries) {
    const std::pair<int, hipEvent_t> cand = P.free_slots.front();
    P.free_slots.pop_front();
    if (cand.second && hipEventQuery(cand.second) == hipErrorNotReady) {
      P.free_slots.push_back(cand); // its last owner's kernels are still running: not yet
      continue;
    }
    (void)hipGetLastError();
    if (cand.second) (void)hipEventDestroy(cand.second);
    slot = cand.first;
  }
  if (slot < 0) return -1; // (this plan runs unguarded)
  __atomic_store_n(&P.h_flags[slot], 0, __ATOMIC_RELAXED);
  *d_guard = P.d_guard + static_cast<size_t>(slot) * kGuardSamples;
  *h_flag = P.h_flags + slot;
  return slot;
}
// `launched`: the plan has enqueued kernels, the last of them on `last_stream` (a stream of `device`)
void guard_release(int device, int slot, bool launched, hipStream_t last_stream) {
  if (slot < 0) return;
  hipEvent_t ev = nullptr;
  if (launched) {
    int cur = -1;
    const bool switched = hipGetDevice(&cur) == hipSuccess && cur != device && hipSetDevice(device) == hipSuccess;
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) ev = nullptr;
    if (ev && hipEventRecord(ev, last_stream) != hipSuccess) { // (e.g. the caller has destroyed that stream: its work is done)
      (void)hipEventDestroy(ev);
      ev = nullptr;
    }
    (void)hipGetLastError();
    if (switched) (void)hipSetDevice(cur);
  }
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guard_pools[device].free_slots.emplace_back(slot, ev);
}

enum Family { kFamRowblock = 0, kFamPlus = 1, kFamFlat = 2, kFamVector = 3, kFamilyCount = 4 };

typedef std::tuple<int, const void *, const void *, const void *, int, int> PlanKey;

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
  bool tuning_open = true;         // some per-matrix timing was deferred (or has not been reached yet): later calls may resume it
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
  Col16 col16;                  // opt-in 16-bit column encoding (tunable col16), built on first use
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
    if (col16.base) (void)hipFree(col16.base);
    if (col16.esc_start) (void)hipFree(col16.esc_start);
    if (col16.esc_cols) (void)hipFree(col16.esc_cols);
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
  // per slab: one entry per run (or piece of a long run): its row, its first non-zero, its place in the pass's virtual non-zero
  // order (entries + 1 prefix sums of the lengths); and the first entry of every workgroup (blocks + 1)
  std::vector<int *> seg_row, seg_begin, seg_vptr, seg_blk;
  std::vector<int> seg_entries, seg_blocks, seg_pieces; // seg_pieces[s] != 0: the slab holds runs cut into pieces (merge kernel needed)
  double *d_seg_ys = nullptr; // one partial sum per entry of the longest list
  int seg_rest_below = 0;      // two-class form: rows of fewer non-zeros than this are whole runs in the last plane (0: every row is cut by slab)
  void free_segments() {
    seg_rest_below = 0;
    for (auto *list : {&seg_row, &seg_begin, &seg_vptr, &seg_blk}) {
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

// ---- tune cache: the per-matrix timed choices, kept across processes --------------------------------------------------------
// The first call on a matrix times a handful of choices (13 ms on the headline matrix, 85 SpMVs' worth) and every process pays
// again; the reference's choice is a pure function of its inputs (strategy_picker.cpp:19-65) and costs nothing.  With
// SPMV_ACC_TUNE_CACHE=<file> (or spmv_acc_set_tune_cache) the choices are appended to a text file, one line per matrix, keyed by a
// digest of (library version, device name, m, n, nnz, the 64 rowptr samples of the stale-plan guard); a later process that meets
// the same matrix on the same device adopts them and only runs the structural passes.  Opt-in; the last line for a key wins;
// a choice the current build cannot honour (a cut-row form that is not legal on this matrix) falls back to the safe one.
struct TuneRecord {
  int v[22]; // stream_policy[4][2], adaptive_family[2], flat_npt, flat_early, flat_geometry_tuned, flat_mode[2], plus_min, hint_state0, hint_use[3], flat_rowblock, seg_choice
  bool operator==(const TuneRecord &o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
};
constexpr int kTuneFields = 22;
std::mutex g_tune_mu;
std::string g_tune_path;
bool g_tune_path_set = false, g_tune_loaded = false;
std::map<unsigned long long, TuneRecord> g_tune_db;

void tune_load_locked() {
  if (!g_tune_path_set) {
    if (const char *e = std::getenv("SPMV_ACC_TUNE_CACHE")) g_tune_path = e;
    g_tune_path_set = true;
  }
  if (g_tune_loaded || g_tune_path.empty()) return;
  g_tune_loaded = true;
  if (FILE *f = std::fopen(g_tune_path.c_str(), "r")) {
    // line by line: a line cut short (a writer killed mid-write) or written by another version is skipped by itself
    char line[1024];
    while (std::fgets(line, sizeof(line), f)) {
      char tag[32];
      unsigned long long key = 0;
      int used = 0;
      if (std::sscanf(line, "%31s %llx%n", tag, &key, &used) != 2 || std::strcmp(tag, "spmvacc3") != 0) continue;
      TuneRecord r;
      bool ok = true;
      const char *at = line + used;
      for (int i = 0; i < kTuneFields && ok; ++i) {
        int step = 0;
        ok = std::sscanf(at, "%d%n", &r.v[i], &step) == 1;
        at += step;
      }
      if (ok && std::strchr(at, '\n')) g_tune_db[key] = r;
    }
    std::fclose(f);
  }
}
bool tune_cache_enabled() {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  tune_load_locked();
  return !g_tune_path.empty();
}
unsigned long long tune_key_of(int dev, int m, int n, int nnz, const int *samples) {
  unsigned long long h = 1469598103934665603ULL; // FNV-1a
  auto mix = [&h](const void *p, size_t bytes) {
    const unsigned char *c = static_cast<const unsigned char *>(p);
    for (size_t i = 0; i < bytes; ++i) h = (h ^ c[i]) * 1099511628211ULL;
  };
  static const char kVersion[] = "spmv_acc_amd 0.3 tune v3";
  mix(kVersion, sizeof(kVersion));
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) == hipSuccess) {
    mix(prop.name, strnlen(prop.name, sizeof(prop.name)));
    mix(prop.gcnArchName, strnlen(prop.gcnArchName, sizeof(prop.gcnArchName)));
    mix(&prop.multiProcessorCount, sizeof(int));
  }
  (void)hipGetLastError();
  mix(&m, sizeof(int));
  mix(&n, sizeof(int));
  mix(&nnz, sizeof(int));
  mix(samples, sizeof(int) * kGuardSamples);
  return h ? h : 1;
}

void tune_log(const char *fmt, ...);
TuneRecord tune_snapshot(const Plan &p) {
  TuneRecord r;
  int k = 0;
  for (int f = 0; f < kFamilyCount; ++f)
    for (int c = 0; c < 2; ++c) r.v[k++] = p.stream_policy[f][c];
  r.v[k++] = p.adaptive_family[0];
  r.v[k++] = p.adaptive_family[1];
  r.v[k++] = p.flat_npt_choice;
  r.v[k++] = p.flat_early_choice ? 1 : 0;
  r.v[k++] = p.flat_geometry_tuned ? 1 : 0;
  r.v[k++] = p.flat_mode_choice[0];
  r.v[k++] = p.flat_mode_choice[1];
  r.v[k++] = p.plus_tuned_min;
  r.v[k++] = p.hint_state == 0 ? 0 : -1; // only "the census found nothing to protect" is worth keeping: the bits themselves are rebuilt
  for (int f = 0; f < 3; ++f) r.v[k++] = p.hint_use[f];
  r.v[k++] = p.flat_rowblock_choice;
  r.v[k++] = p.seg_choice;
  return r;
}
} // namespace
void set_tune_cache(const char *path) {
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_tune_path = path ? path : "";
  g_tune_path_set = true;
  g_tune_loaded = false;
  g_tune_db.clear();
}
namespace {
// a fresh plan adopts what an earlier process (or an earlier plan of this process) measured on the same matrix
void tune_adopt(Plan &p) {
  TuneRecord r;
  {
    std::lock_guard<std::mutex> lk(g_tune_mu);
    auto it = g_tune_db.find(p.tune_key);
    if (it == g_tune_db.end()) return;
    r = it->second;
  }
  auto in = [](int v, int lo, int hi) { return v >= lo && v <= hi; };
  int k = 0;
  for (int f = 0; f < kFamilyCount; ++f)
    for (int c = 0; c < 2; ++c, ++k) p.stream_policy[f][c] = in(r.v[k], 0, 3) ? r.v[k] : -1;
  p.adaptive_family[0] = in(r.v[k], 0, 2) ? r.v[k] : -1;
  ++k;
  p.adaptive_family[1] = in(r.v[k], 0, 2) ? r.v[k] : -1;
  ++k;
  p.flat_npt_choice = (r.v[k] == 4 || r.v[k] == 8) ? r.v[k] : 0;
  ++k;
  p.flat_early_choice = r.v[k++] == 1;
  p.flat_geometry_tuned = r.v[k++] == 1 && p.flat_npt_choice > 0;
  p.flat_mode_choice[0] = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.flat_mode_choice[1] = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.plus_tuned_min = (r.v[k] == 1024 || r.v[k] == 1536 || r.v[k] == 1920) ? r.v[k] : 0;
  ++k;
  if (r.v[k++] == 0) p.hint_state = 0;
  for (int f = 0; f < 3; ++f, ++k) p.hint_use[f] = in(r.v[k], 0, 1) ? r.v[k] : -1;
  p.flat_rowblock_choice = in(r.v[k], 0, 1) ? r.v[k] : -1;
  ++k;
  p.seg_choice = in(r.v[k], 0, 1) ? r.v[k] : -1;
  tune_log("m %d nnz %d: choices adopted from the tune cache (key %016llx)", p.A.m, p.A.nnz, p.tune_key);
}
// after a call that did plan work: keep what the plan now knows
void tune_store(const Plan &p) {
  if (!p.tune_key) return;
  const TuneRecord r = tune_snapshot(p);
  std::lock_guard<std::mutex> lk(g_tune_mu);
  if (g_tune_path.empty()) return;
  auto it = g_tune_db.find(p.tune_key);
  if (it != g_tune_db.end() && it->second == r) return;
  g_tune_db[p.tune_key] = r;
  if (FILE *f = std::fopen(g_tune_path.c_str(), "a")) { // one line, one write: concurrent processes interleave whole lines
    std::string line = "spmvacc3 ";
    char buf[32];
    std::snprintf(buf, sizeof(buf), "%016llx", p.tune_key);
    line += buf;
    for (int i = 0; i < kTuneFields; ++i) line += " " + std::to_string(r.v[i]);
    line += "\n";
    std::fwrite(line.data(), 1, line.size(), f);
    std::fclose(f);
  }
}

// Derived matrices (the slabs of the opt-in column-slab blocking) have plans of their own, keyed by pointers into their parent's
// arrays.  When the parent dies those plans must go too; a plan can die under g_mu, so the rowptrs are queued here and the entries
// are erased the next time the cache is touched (before any lookup: a re-used address never meets a dead plan).
std::mutex g_deferred_mu;
std::vector<const void *> g_deferred_rp;
void Plan::free_slabs() {
  if (!slab_crp.empty()) {
    std::lock_guard<std::mutex> lk(g_deferred_mu);
    for (int *crp : slab_crp)
      if (crp) g_deferred_rp.push_back(crp); // (the slabs' plans are keyed by their compact row pointers)
  }
  for (int *q : slab_rowid)
    if (q) (void)hipFree(q);
  for (int *q : slab_crp)
    if (q) (void)hipFree(q);
  slab_rowid.clear();
  slab_crp.clear();
  slab_rows.clear();
  if (d_slab_ys) (void)hipFree(d_slab_ys);
  d_slab_ys = nullptr;
  if (d_slab_rp) (void)hipFree(d_slab_rp);
  if (d_slab_ci) (void)hipFree(d_slab_ci);
  if (d_slab_v) (void)hipFree(d_slab_v);
  if (d_slab_off) (void)hipFree(d_slab_off);
  d_slab_off = nullptr;
  d_slab_rp = d_slab_ci = nullptr;
  d_slab_v = nullptr;
  slab_count = 0;
  slab_off.clear();
}

std::map<PlanKey, std::shared_ptr<Plan>> g_plans; // a running call keeps its plan alive through its own reference
void drain_deferred_locked() { // g_mu held
  std::vector<const void *> dead;
  {
    std::lock_guard<std::mutex> lk(g_deferred_mu);
    dead.swap(g_deferred_rp);
  }
  while (!dead.empty()) { // (erasing a plan may queue more)
    for (auto it = g_plans.begin(); it != g_plans.end();) {
      if (std::find(dead.begin(), dead.end(), std::get<1>(it->first)) != dead.end()) it = g_plans.erase(it);
      else ++it;
    }
    dead.clear();
    std::lock_guard<std::mutex> lk(g_deferred_mu);
    dead.swap(g_deferred_rp);
  }
}
thread_local std::weak_ptr<Plan> t_last_plan;     // the plan this thread's latest run_spmv used (last_error asks it, and only it)

// Is the calling thread inside a stream capture (set by run_spmv)?  Plan work -- allocations, synchronisation, timings -- would
// invalidate the capture: required work is refused with an error that says so, optional work (timed choices) is skipped and the
// call runs with what the plan already holds.
thread_local bool t_capturing = false;
bool plan_work_allowed(const char *what) {
  if (!t_capturing) return true;
  set_error(kErrBadArgument, std::string("this call needs plan work (") + what +
                                 ") that allocates or synchronises and cannot run inside a stream capture: run the same call "
                                 "(or spmv_acc_prepare with this strategy) once outside the capture first; nothing was enqueued");
  return false;
}
constexpr size_t kMaxPlans = 1024; // beyond this the least recently used plan is dropped
unsigned long long g_use_clock = 0;

// Is p readable by the host?  The reference's sparse_spmv hands the SAME device pointer in as "host"
// rowptr (api/spmv_imp.cpp:14-17), which only works with host-visible device memory.
bool host_readable(const void *p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  std::memset(&attr, 0, sizeof(attr));
  const hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError(); // plain malloc'ed memory is not known to HIP: that is a host pointer
    return true;
  }
  return attr.type == hipMemoryTypeHost || attr.type == hipMemoryTypeUnregistered ||
         attr.type == hipMemoryTypeManaged;
}

// h if the host may dereference it, else null (checked only on the once-per-matrix paths)
const int *host_view(const int *h) { return host_readable(h) ? h : nullptr; }

bool fetch_samples(Plan &p, const int *h_rowptr) {
  if (p.have_samples) return true;
  if (!plan_work_allowed("reading the rowptr samples")) return false;
  ++t_plan_work;
  h_rowptr = host_view(h_rowptr);
  const int m = p.A.m;
  const int idx[4] = {m / 4, m / 2, static_cast<int>(3LL * m / 4), m};
  int out[4];
  if (h_rowptr) {
    for (int i = 0; i < 4; ++i) out[i] = h_rowptr[idx[i]];
  } else {
    for (int i = 0; i < 4; ++i) {
      if (!hip_ok(hipMemcpy(&out[i], p.A.rp + idx[i], sizeof(int), hipMemcpyDeviceToHost), "read rowptr sample"))
        return false;
    }
  }
  p.samples.q1 = out[0];
  p.samples.half = out[1];
  p.samples.q3 = out[2];
  p.samples.last = out[3];
  p.have_samples = true;
  return true;
}

std::shared_ptr<Plan> get_plan(int m, int n, int nnz, const int *h_rowptr, const int *rp, const int *ci, const double *v) {
  int dev = 0;
  if (!hip_ok(hipGetDevice(&dev), "hipGetDevice")) return nullptr;
  const PlanKey key(dev, rp, ci, v, m, n);
  std::lock_guard<std::mutex> lk(g_mu);
  drain_deferred_locked();
  auto it = g_plans.find(key);
  if (it != g_plans.end() && it->second->is_stale()) {
    if (!plan_work_allowed("rebuilding a stale plan")) return nullptr;
    set_error(kErrBadArgument,
              "the matrix behind a cached plan changed (same pointers and shape, different rowptr) without "
              "spmv_acc_release_plans: the results of the EARLIER calls made on it since the change are invalid; the plan has "
              "been rebuilt, so the call that reports this ran on the matrix as it is now and its y is valid");
    g_plans.erase(it);
    it = g_plans.end();
  }
  if (it != g_plans.end()) {
    if (nnz < 0 || nnz == it->second->A.nnz) {
      it->second->last_use = ++g_use_clock;
      return it->second;
    }
    // same buffers, different nnz: the caller rebuilt the matrix in place
    if (!plan_work_allowed("rebuilding the plan of a matrix whose nnz changed")) return nullptr;
    g_plans.erase(it);
  }
  if (!plan_work_allowed("building the plan of a matrix seen for the first time")) return nullptr;
  if (g_plans.size() >= kMaxPlans) {
    auto oldest = g_plans.begin();
    for (auto jt = g_plans.begin(); jt != g_plans.end(); ++jt)
      if (jt->second->last_use < oldest->second->last_use) oldest = jt;
    g_plans.erase(oldest);
  }
  if (nnz < 0) {
    if ((h_rowptr = host_view(h_rowptr)) != nullptr) {
      nnz = h_rowptr[m];
    } else if (!hip_ok(hipMemcpy(&nnz, rp + m, sizeof(int), hipMemcpyDeviceToHost), "read rowptr[m]")) {
      return nullptr;
    }
  }
  if (nnz < 0 || nnz > INT_MAX - (1 << 16)) {
    set_error(kErrTooLarge, "nnz does not leave room for tile arithmetic in int32; shard the matrix");
    return nullptr;
  }
  std::shared_ptr<Plan> p = std::make_shared<Plan>();
  p->device = dev;
  p->key = key;
  p->A.m = m;
  p->A.n = n;
  p->A.nnz = nnz;
  p->A.rp = rp;
  p->A.ci = ci;
  p->A.v = v;
  p->A.aligned16 = (reinterpret_cast<uintptr_t>(ci) % 16 == 0) && (reinterpret_cast<uintptr_t>(v) % 16 == 0) &&
                   nnz >= 8;
  p->last_use = ++g_use_clock;
  p->guard_slot = guard_acquire(dev, &p->A.guard, &p->A.stale);
  if (p->guard_slot >= 0) {
    launch_guard_fill(t_stream, rp, m, const_cast<int *>(p->A.guard));
    if (!hip_ok(hipStreamSynchronize(t_stream), "record the plan guard")) return nullptr; // (a later call may use another stream)
    if (tune_cache_enabled() && !tun(kT_deterministic)) {
      int samples[kGuardSamples];
      if (hipMemcpy(samples, p->A.guard, sizeof(samples), hipMemcpyDeviceToHost) == hipSuccess) {
        p->tune_key = tune_key_of(dev, m, n, nnz, samples);
        tune_adopt(*p);
      }
      (void)hipGetLastError();
    }
  }
  g_plans[key] = p;
  return p;
}

const char *const kStaleText =
    "the matrix behind a cached plan changed (same pointers and shape, different rowptr) without spmv_acc_release_plans: the "
    "results of the calls made on it since the change -- including the calling thread's most recent SpMV on these pointers -- "
    "are invalid; the plan has been dropped and the next call on the matrix rebuilds it";

// O(1): the plan the calling thread used last, nothing else
bool report_stale_last_plan() {
  const std::shared_ptr<Plan> p = t_last_plan.lock();
  if (!p || !p->is_stale()) return false;
  set_error(kErrBadArgument, kStaleText);
  t_last_plan.reset();
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_plans.find(p->key);
  if (it != g_plans.end() && it->second == p) g_plans.erase(it);
  return true;
}
} // namespace

// Every cached plan (any thread's): drops the stale ones, returns how many there were.  For callers that edit matrices in place
// from several threads and want one check after a device-wide synchronisation; not on any hot path.
int check_plans() {
  std::lock_guard<std::mutex> lk(g_mu);
  int dropped = 0;
  for (auto it = g_plans.begin(); it != g_plans.end();) {
    if (it->second->is_stale()) {
      it = g_plans.erase(it);
      ++dropped;
    } else {
      ++it;
    }
  }
  if (dropped) set_error(kErrBadArgument, kStaleText);
  return dropped;
}

namespace {

// Break points, carry buffers and the two plan-time probes of a flat plan with `stride` non-zeros per tile.
bool build_flat_plan(const CsrDev &A, int stride, hipStream_t stream, FlatPlan &F) {
  if (!plan_work_allowed("flat: break points and tile digests")) return false;
  ++t_plan_work;
  Plan::free_flat_plan(F);
  const int nnz = A.nnz;
  const int tiles = nnz / stride + (nnz % stride ? 1 : 0);
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
  launch_break_points(stream, A.rp, A.m, nnz, stride, F.bp, static_cast<int>(n1));
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
constexpr int kFlatSmallNnz = 24 << 20;

thread_local bool t_flat_segment_sum = false; // this thread is inside segment_sum_flat_sparse_spmv (FlatSegmentSumScope)
inline bool flat_segment_sum() { return (t_flat_segment_sum || tun(kT_flat_reduce) == 1) && tun(kT_col16) <= 0; }

int flat_stride_for(const Plan &p) {
  if (flat_segment_sum()) return kThreads * kNnzPerThread; // the scan is written for the 2048-non-zero tile
  if (tun(kT_col16) > 0) return kThreads * kNnzPerThread; // the 16-bit encoding is read by the 2048-non-zero tile
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

} // namespace

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

namespace {

bool ensure_plus(Plan &p, const int *h_rowptr, hipStream_t stream, int min_nnz) {
  if (min_nnz < 256 || min_nnz > kTile) min_nnz = kPlusMinNnz;
  const int want_vec =
      tun(kT_plus_ref_vec) ? plus_pick_vec(p.A.m, p.A.nnz) : plus_pick_vec_tuned(p.A.m, p.A.nnz, min_nnz);
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
inline double trial_beta() { return t_beta_class ? 1.0 : 0.0; }

int policy_for(const Plan &p, int fam) {
  const int forced = tun(kT_stream_plain);
  if (forced >= 0) return forced & 3;
  // deterministic: the rule the timings follow on most matrices -- short rows (the vectors and rowptr are worth more cache than
  // the matrix) stream non-temporally, everything else with the default policy
  if (tun(kT_deterministic)) return static_cast<long long>(p.A.nnz) <= 8LL * p.A.m ? kStreamPolicyNt : kStreamPolicyDefault;
  const int c = t_beta_class;
  if (p.stream_policy[fam][c] >= 0) return p.stream_policy[fam][c];
  // not timed for this family in this class yet (adaptive's comparison of the families): the policy another family measured on
  // this matrix in the same class is a far better guess than a fixed one, then this family's other class
  for (int f = 0; f < kFamilyCount; ++f)
    if (p.stream_policy[f][c] >= 0) return p.stream_policy[f][c];
  if (p.stream_policy[fam][c ^ 1] >= 0) return p.stream_policy[fam][c ^ 1];
  return static_cast<long long>(p.A.nnz) <= 8LL * p.A.m ? kStreamPolicyNt : kStreamPolicyDefault; // (nothing measured yet: the rule)
}

// While adaptive compares the families it runs each with its default sub-choices (flat: carries + fix-up unless pinned;
// row-block-plus: MIN_NNZ 1536); the family that wins refines its own sub-choice on its next call.
thread_local bool t_coarse_tuning = false;
thread_local bool t_no_policy_timing = false; // run_plus's early slab decision: the row-block-plus kernel runs under the rule's cache policy, nothing is timed for it

// ---- plan-time budget (tunables first_call_budget / later_call_budget) ---------------------------------------------------------------
// The reference pays a fixed, small preprocessing cost per call (hip-flat/flat.cpp:39-44: one malloc + memset + break-point kernel); a plan
// that spends 64 SpMVs' worth of trial launches on its first call gives that advantage back to short solves.  So the trial launches of
// a call are bounded: run_spmv notes when the call began and how many SpMV-equivalents it may spend; the first trial launch measured in the
// call (TuneTimer) turns that into milliseconds; every timing PHASE asks defer_tuning() before it starts and, when the budget is
// spent, leaves its choice open (the `deterministic` rule serves the call) for a later call to settle.  Structural passes are not
// deferred -- a call cannot run without them -- but their time counts as spent.
thread_local std::chrono::steady_clock::time_point t_call_began;
thread_local double t_budget_spmvs = 0.0; // SpMV-equivalents this call may spend; <= 0: unbounded
thread_local double t_budget_ms = -1.0;   // the same in milliseconds, known once a trial launch has been measured in this call (or from the plan)
thread_local bool t_tuning_deferred = false; // some phase of this call left its choice open
thread_local float t_first_trial_ms = 0.f;   // the first trial launch this call measured (0: none)
thread_local int t_unbounded_tuning = 0;     // > 0: this thread is inside spmv_acc_prepare (UnboundedTuningScope): no budget
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
  // The trial launches write a scratch y.  With a reset buffer set (and tunable tune_protocol 1, the default) the timed launches
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
        if (reset_ptr && tun(kT_tune_protocol) == 1) (void)hipMemsetAsync(reset_ptr, 0, reset_bytes, st);
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
    const int timed = first < 0.1f ? 5 : (first < 0.5f ? (lean ? 2 : 3) : (first < 2.0f ? 2 : 1));
    for (int w = 0; w < warm; ++w) fn();
    if (reset_ptr && tun(kT_tune_protocol) == 1) {
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
};

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

// Opt-in 16-bit column encoding of the whole matrix (k_col16.hip): base + escape count per 256-non-zero chunk, exclusive scan
// of the counts, then the offsets and the escape list.  One synchronisation (the escape total sizes the last allocation).
bool ensure_col16(Plan &p, hipStream_t st) {
  constexpr size_t kWarpPad = 64;
  if (p.col16.d16) return true;
  if (!plan_work_allowed("the 16-bit column encoding")) return false;
  ++t_plan_work;
  Col16 &C = p.col16;
  const int nchunks = (p.A.nnz + kCol16Chunk - 1) / kCol16Chunk;
  int *esc_count = nullptr;
  void *tmp = nullptr;
  const size_t tmp_bytes = col16_scan_bytes(nchunks);
  bool ok = hip_ok(hipMalloc(reinterpret_cast<void **>(&C.d16), sizeof(unsigned short) * static_cast<size_t>(nchunks) * kCol16Chunk),
                   "hipMalloc col16 offsets") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&C.base), sizeof(int) * static_cast<size_t>(nchunks)), "hipMalloc col16 bases") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&C.esc_start), sizeof(int) * (static_cast<size_t>(nchunks) + 1)),
                   "hipMalloc col16 escape offsets") &&
            hip_ok(hipMalloc(reinterpret_cast<void **>(&esc_count), sizeof(int) * (static_cast<size_t>(nchunks) + 1)),
                   "hipMalloc col16 escape counts") &&
            hip_ok(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 16), "hipMalloc col16 scan workspace") &&
            hip_ok(hipMemsetAsync(esc_count + nchunks, 0, sizeof(int), st), "memset col16");
  int total = 0;
  if (ok) {
    launch_col16_base(st, p.A.ci, p.A.nnz, nchunks, C.base, esc_count);
    ok = launch_col16_scan(st, nchunks, esc_count, C.esc_start, tmp, tmp_bytes) &&
         hip_ok(hipMemcpyAsync(&total, C.esc_start + nchunks, sizeof(int), hipMemcpyDeviceToHost, st), "read col16 escape total") &&
         hip_ok(hipStreamSynchronize(st), "sync col16") &&
         // (+ 64: every wavefront preloads 64 entries from its chunk's first escape on, also at the very end of the list)
         hip_ok(hipMalloc(reinterpret_cast<void **>(&C.esc_cols), sizeof(int) * (static_cast<size_t>(total) + kWarpPad)), "hipMalloc col16 escapes") &&
         hip_ok(hipMemsetAsync(C.esc_cols, 0, sizeof(int) * (static_cast<size_t>(total) + kWarpPad), st), "memset col16 escapes");
  }
  if (ok) {
    launch_col16_encode(st, p.A.ci, p.A.nnz, nchunks, C.base, C.esc_start, C.d16, C.esc_cols);
    ok = hip_ok(hipStreamSynchronize(st), "sync col16 encode");
  }
  if (esc_count) (void)hipFree(esc_count);
  if (tmp) (void)hipFree(tmp);
  if (!ok) {
    p.free_col16();
    return false;
  }
  C.nchunks = nchunks;
  C.escapes = total;
  return true;
}

// walking direction of this plan's next tile-kernel launch (tunable zigzag): consecutive SpMVs on a matrix alternate
inline bool next_reverse(Plan &p) { return tun(kT_zigzag) && (p.launches++ & 1u); }

void launch_flat_plan(hipStream_t st, const CsrDev &A, FlatPlan &F, int policy, double alpha, double beta, const double *x,
                      double *y, bool reverse) {
  F.xcd_chunk = tun(kT_stage_fast) ? tun(kT_xcd_chunk_tiles) : -1; // -1: per-lane predicated staging (A/B)
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
  if (t_coarse_tuning && p.A.nnz >= kFlatSmallNnz) { // (small matrices: the timings are cheap and decide the comparison)
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
  if (p.A.nnz >= kFlatSmallNnz || p.flat.ntiles <= 1) {
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
  if (mode == 0 || A.nnz < 8 || A.n <= 0 || static_cast<long long>(A.n) * 8 >= (1LL << 32)) return true;
  // (and while x lives in the 256 MB Infinity Cache beside the rest of the working set a cold gather is a hit there, which a non-temporal
  // load forfeits: R-MAT scale 21 / 22 / 23, x = 16 / 32 / 64 MB: hinted 268 / 604 / 1343 us against 212 / 477 / 1250 plain; scale 24 / 25,
  // x = 128 / 256 MB: 3.03 / 7.2 ms against 3.38 / 8.2 -- the timed choice gets all five right, this bound just saves the census)
  if (mode < 0 && static_cast<long long>(A.n) * 8 < (96LL << 20)) return true;
  ++t_plan_work;
  const auto census_t0 = std::chrono::steady_clock::now();
  const int nlines = (A.n + (1 << kHintLineShift) - 1) >> kHintLineShift;
  // (odd: an even stride on rows of one even length would sample the same position of every row -- with sorted rows always low columns)
  const int stride = ((A.nnz + kHintSamples - 1) / kHintSamples) | 1;
  const int samples = (A.nnz + stride - 1) / stride;
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
    launch_hint_census(st, A.ci, A.nnz, A.n, stride, samples, counts);
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
        launch_hint_bits(st, A.ci, A.nnz, A.n, counts, threshold, p.d_cold);
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

bool run_rowblock(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y,
                  bool allow_uneven_switch = false, int lanes_per_row = 0);
bool probe_rowblock(Plan &p, int rpb, hipStream_t st);

// A flat tile is one workgroup and walks its rows 256 at a time.  Where a tile owns tens of thousands of rows (hypersparse
// matrices: 50 M rows with 6000 non-zeros put all of them into ONE tile, 96 ms) the rows, not the non-zeros, need cutting: such
// matrices run the fixed row blocks instead (0.27 ms) -- the mirror image of the row-block family's rescue.
constexpr int kFlatMaxTileRows = 16384;

bool run_flat(hipStream_t st, Plan &p, double alpha, double beta, const double *x, double *y) {
  if (!ensure_flat(p, st)) return false;
  if (p.flat.max_tile_rows > kFlatMaxTileRows && tun(kT_rowblock_guard)) {
    // (guard against mutual recursion: the row-block rescue goes to row-block-plus unless rescue_flat is set, and a matrix with
    // such tiles has no row-block imbalance of the hub-row kind)
    if (!(tun(kT_rescue_flat) && p.rowblock_ok == 0)) return run_rowblock(st, p, nullptr, alpha, beta, x, y, false, 0);
  }
  p.A.cold = nullptr; // (the plan-time timings below run without gather hints)
  p.flat.col16 = nullptr;
  if (tun(kT_col16) > 0) {
    if (!ensure_col16(p, st)) return false;
    p.flat.col16 = &p.col16; // (used by the 2048-non-zero tile only; other tile sizes read colindex)
  }
  if (!autotune_policy(p, kFamFlat, st, [&](int pol, double *ys) { launch_flat_with(st, p, pol, 1.0, trial_beta(), x, ys); })) return false;
  if (!autotune_flat_mode(p, st, x)) return false;
  if (!autotune_flat_geometry(p, st, x)) return false;
  if (!autotune_hint(p, kFamFlat, st, [&](double *ys) { launch_flat_with(st, p, policy_for(p, kFamFlat), 1.0, trial_beta(), x, ys); })) return false;
  // Small grids: the tile kernel's extra dependent hop (tile digest -> row extents) is not hidden by other workgroups.  Where the
  // fixed row blocks are balanced (nothing for non-zero-cut tiles to repair) the two kernels are timed once and the faster runs.
  const int rb_mode = tun(kT_flat_rowblock);
  // (a caller that pins any of the tile kernel's own choices -- cut-row form, tile size, staging order -- is asking for that kernel)
  const bool tile_pinned = tun(kT_flat_finish) >= 0 || tun(kT_flat_npt) >= 0 || tun(kT_flat_early) >= 0;
  if (rb_mode != 0 && (rb_mode > 0 || !tile_pinned) && !flat_segment_sum() && tun(kT_col16) <= 0 &&
      !tun(kT_rescue_flat) && !t_coarse_tuning) {
    if (rb_mode > 0) {
      int vec = 1, rpb = kThreads;
      pick_rowblock_shape(p.A.m, p.A.nnz, tun(kT_rowblock_target), &vec, &rpb);
      if (t_capturing ? (p.rowblock_ok == 1 && p.rowblock_rpb == rpb) : (probe_rowblock(p, rpb, st) && p.rowblock_ok == 1))
        return run_rowblock(st, p, nullptr, alpha, beta, x, y, false, 0);
    } else if (p.flat_rowblock_choice < 0 && !t_capturing && !by_rule()) {
      int vec = 1, rpb = kThreads;
      pick_rowblock_shape(p.A.m, p.A.nnz, tun(kT_rowblock_target), &vec, &rpb);
      if (!probe_rowblock(p, rpb, st)) return false;
      p.flat_rowblock_choice = 0;
      if (p.rowblock_ok == 1) {
        ++t_plan_work;
        double *scratch = nullptr;
        if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
        TuneTimer timer;
        timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
        bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
        float ms_flat = 0.f, ms_rb = 0.f;
        ok = ok && run_rowblock(st, p, nullptr, 1.0, trial_beta(), x, scratch, false); // (builds and tunes the row-block side)
        ok = ok && timer.time(st, [&] { launch_flat_with(st, p, policy_for(p, kFamFlat), 1.0, trial_beta(), x, scratch); }, &ms_flat);
        ok = ok && timer.time(st, [&] { (void)run_rowblock(st, p, nullptr, 1.0, trial_beta(), x, scratch, false); }, &ms_rb);
        if (!ok) return false;
        p.flat_rowblock_choice = ms_rb < 0.97f * ms_flat ? 1 : 0;
        tune_log("m %d nnz %d flat on balanced rows: tile kernel %.2f us, row blocks %.2f us -> %s", p.A.m, p.A.nnz, ms_flat * 1e3f, ms_rb * 1e3f,
                 p.flat_rowblock_choice ? "row blocks" : "tile kernel");
      }
    }
    if (rb_mode < 0 && p.flat_rowblock_choice == 1) return run_rowblock(st, p, nullptr, alpha, beta, x, y, false, 0);
  }
  launch_flat_with(st, p, policy_for(p, kFamFlat), alpha, beta, x, y);
  return true;
}

// Opt-in (tunable `validate`): one pass over rowptr and colindex per new matrix; a matrix that fails is refused on this
// and every later call (until its plan is released) instead of sending a kernel out of bounds.
bool validate_plan(Plan &p, hipStream_t st) {
  if (p.invalid < 0) {
    if (!plan_work_allowed("validating the matrix")) return false;
    ++t_plan_work;
    int *d_flags = nullptr;
    int h = -1;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d_flags), sizeof(int)), "hipMalloc validate")) return false;
    if (hip_ok(hipMemsetAsync(d_flags, 0, sizeof(int), st), "memset validate")) {
      launch_validate_csr(st, p.A, d_flags);
      if (!hip_ok(hipMemcpyAsync(&h, d_flags, sizeof(int), hipMemcpyDeviceToHost, st), "read validate") ||
          !hip_ok(hipStreamSynchronize(st), "sync validate"))
        h = -1;
    }
    (void)hipFree(d_flags);
    if (h < 0) return false;
    p.invalid = h;
  }
  if (p.invalid != 0) {
    std::string what = "matrix failed validation:";
    if (p.invalid & 1) what += " rowptr decreases or is negative;";
    if (p.invalid & 2) what += " rowptr[m] != nnz;";
    if (p.invalid & 4) what += " column index outside [0, n);";
    set_error(kErrBadArgument, what);
    return false;
  }
  return true;
}

// Opt-in (tunable guard_full): the digest of the whole rowptr, taken once per plan; then one digest + verdict pair per call, on the
// call's stream AHEAD of its SpMV kernels (the flag is up by the time the caller has synchronised and asks spmv_acc_last_error).
bool launch_full_guard(Plan &p, hipStream_t st) {
  if (!p.A.guard || !p.A.stale) return true; // (the plan runs unguarded: no slot was free)
  if (!p.have_rp_digest) {
    if (!plan_work_allowed("the full row-pointer digest (guard_full)")) return false;
    ++t_plan_work;
    // kDigestSlots sets of partial sums + one word for the build-time digest
    const size_t words = static_cast<size_t>(Plan::kDigestSlots) * kDigestMaxParts + 1;
    if (!p.d_digest_acc && !hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_digest_acc), sizeof(unsigned long long) * words), "hipMalloc digest"))
      return false;
    unsigned long long *out = p.d_digest_acc + words - 1;
    launch_rowptr_digest(st, p.A.rp, p.A.m, p.d_digest_acc);
    launch_rowptr_verdict(st, p.d_digest_acc, p.A.m, 0, nullptr, out);
    unsigned long long h = 0;
    if (!hip_ok(hipMemcpyAsync(&h, out, sizeof(h), hipMemcpyDeviceToHost, st), "read digest") || !hip_ok(hipStreamSynchronize(st), "sync digest"))
      return false;
    p.rp_digest = h;
    p.have_rp_digest = true;
  }
  unsigned long long *part = p.d_digest_acc + static_cast<size_t>(p.digest_turn++ % Plan::kDigestSlots) * kDigestMaxParts;
  launch_rowptr_digest(st, p.A.rp, p.A.m, part);
  launch_rowptr_verdict(st, part, p.A.m, p.rp_digest, p.A.stale, nullptr);
  return true;
}

// Once per matrix: would any fixed row block have to stream more than kRowblockMaxRounds tiles?
// (power-law matrices: R-MAT hub rows put millions of non-zeros into one workgroup.)
bool probe_rowblock(Plan &p, int rpb, hipStream_t st) {
  if (p.rowblock_ok >= 0 && p.rowblock_rpb == rpb) return true;
  if (!plan_work_allowed("the row-block balance probe")) return false;
  ++t_plan_work;
  p.rowblock_rpb = rpb;
  int *d_max = nullptr;
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d_max), 2 * sizeof(int)), "hipMalloc probe")) return false;
  bool ok = hip_ok(hipMemsetAsync(d_max, 0, 2 * sizeof(int), st), "memset probe");
  if (ok) {
    const long long nblocks = (static_cast<long long>(p.A.m) + rpb - 1) / rpb;
    const long long avg_block = nblocks > 0 ? p.A.nnz / nblocks : 0;
    launch_max_block_nnz(st, p.A.rp, p.A.m, rpb, static_cast<int>(avg_block), d_max);
    int h[2] = {0, 0};
    ok = hip_ok(hipMemcpyAsync(h, d_max, 2 * sizeof(int), hipMemcpyDeviceToHost, st), "read probe") &&
         hip_ok(hipStreamSynchronize(st), "sync probe");
    if (ok) {
      p.max_block_nnz = h[0];
      // balanced enough = the heaviest block needs few LDS rounds AND is not far above the average block
      // (a block that fits one tile is always fine)
      const bool few_rounds = h[0] <= kRowblockMaxRounds * kTile;
      const bool near_avg = h[0] <= kTile || h[0] <= 4 * avg_block;
      p.rowblock_ok = (few_rounds && near_avg) ? 1 : 0;
      // uneven = a fifth or more of the blocks are > 35 % away from the average block or spill into a second LDS round: fixed
      // row blocks then alternate between half-empty tiles and second rounds (striped densities), and blocks cut by
      // non-zero count do better
      p.rowblock_uneven = nblocks >= 16 && 5LL * h[1] >= nblocks;
    }
  }
  (void)hipFree(d_max);
  return ok;
}

// Row digest of the row-block family (kernels.hpp RowDigest): derived from rowptr alone, like everything else a plan holds.
bool ensure_digest(Plan &p, int rpb, hipStream_t st) {
  if (p.digest.lens && p.digest.rpb == rpb) return true;
  if (!plan_work_allowed("the row digest")) return false;
  ++t_plan_work;
  p.free_digest();
  const size_t nblocks = (static_cast<size_t>(p.A.m) + rpb - 1) / rpb;
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&p.digest.lens), static_cast<size_t>(p.A.m)), "hipMalloc row lengths") ||
      !hip_ok(hipMalloc(reinterpret_cast<void **>(&p.digest.base), sizeof(int) * (nblocks + 1)), "hipMalloc row-block bases")) {
    p.free_digest();
    return false;
  }
  launch_row_digest(st, p.A.rp, p.A.m, rpb, p.digest.lens, p.digest.base);
  if (!hip_ok(hipStreamSynchronize(st), "build the row digest")) { // (a later call may run on another stream)
    p.free_digest();
    return false;
  }
  p.digest.rpb = rpb;
  return true;
}

// line-enhance family with its imbalance rescue: fixed row blocks while every block stays within a
// few LDS rounds, otherwise the same tile machinery cut by non-zeros (flat) so hub rows are shared
// by many workgroups instead of serialising one.
bool run_plus(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y);

bool run_rowblock(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y,
                  bool allow_uneven_switch, int lanes_per_row) {
  int vec = 1, rpb = kThreads;
  pick_rowblock_shape(p.A.m, p.A.nnz, tun(kT_rowblock_target), &vec, &rpb);
  // (THREAD_ROW: one lane per row whatever the row length, the rows per workgroup still from the tile target)
  if (lanes_per_row > 0) vec = lanes_per_row;
  const int forced = tun(kT_rowblock_vec);
  if (forced > 0) {
    vec = forced;
    rpb = kThreads / forced;
  }
  if (tun(kT_rowblock_guard)) {
    if (!probe_rowblock(p, rpb, st)) return false;
    // Imbalanced (power-law) matrix: fixed row blocks would leave a few workgroups with most of the work.  The rescue
    // is the row-block-PLUS kernel -- the reference's own answer to this (hip-csr-adaptive-plus is its line-enhance
    // kernel over analysed row blocks, long rows cut into dedicated blocks) -- which measures 1 % (R-MAT scale 25),
    // 5 % (scale 22) and 17 % (scale 20) faster than the nnz-cut tiles of flat; `rescue_flat` keeps the older choice.
    if (p.rowblock_ok == 0)
      return tun(kT_rescue_flat) ? run_flat(st, p, alpha, beta, x, y) : run_plus(st, p, h_rowptr, alpha, beta, x, y);
    // Uneven but not pathological (striped densities: 60 / 20 nnz per row alternating every 300 or 5000 rows ran 196 us here and
    // 177 us in row-block-plus; 30 / 10 every 64 rows 108 vs 97 us): same answer, for the strategies that leave the choice to
    // the engine.  line / line-enhance / thread_row keep their fixed row blocks.
    if (p.rowblock_uneven && allow_uneven_switch && !tun(kT_rescue_flat)) return run_plus(st, p, h_rowptr, alpha, beta, x, y);
  }
  const RowDigest *dg = nullptr;
  const int want_lens = tun(kT_rowlen);
  // (auto: rows of <= 8 non-zeros on average, where rowptr is >= 3.5 % of the traffic; measured at 12.6 per row the scan costs
  // more than the bytes save -- largebasis-sized 17.8 vs 17.4 us)
  if (want_lens > 0 || (want_lens < 0 && static_cast<long long>(p.A.nnz) <= 8LL * p.A.m)) {
    // (inside a capture a digest that does not exist yet is simply not used: the kernel reads rowptr, same result)
    const bool have = p.digest.lens && p.digest.rpb == rpb;
    if (have || !t_capturing) {
      if (!ensure_digest(p, rpb, st)) return false;
      dg = &p.digest;
    }
  }
  // blocks at each end of the grid whose streams stay cacheable (tunable cache_ends_mb; 12 B per non-zero of stream)
  int cache_ends = 0;
  if (tun(kT_cache_ends_mb) > 0 && tun(kT_zigzag) && p.A.nnz > 0) {
    const long long nblocks = (static_cast<long long>(p.A.m) + rpb - 1) / rpb;
    const double bytes_per_block = 12.0 * p.A.nnz / static_cast<double>(nblocks);
    cache_ends = static_cast<int>(tun(kT_cache_ends_mb) * 1048576.0 / bytes_per_block);
  }
  const int chunk = tun(kT_xcd_chunk);
  const int base_flags = (tun(kT_xcd_remap) ? 1 : 0) | (tun(kT_early_y) ? 2 : 0) |
                         (chunk > 0 ? (4 | (chunk << 8)) : 0) | (tun(kT_stage_fast) ? 0 : 8);
  p.A.cold = nullptr; // (the policy timing runs without gather hints)
  if (!autotune_policy(p, kFamRowblock, st, [&](int pol, double *ys) {
        const int zz = next_reverse(p) ? 64 : 0;
        launch_rowblock_stream(st, p.A, vec, rpb, base_flags | (pol << 4) | zz, 1.0, trial_beta(), x, ys, dg, cache_ends);
      }))
    return false;
  if (!autotune_hint(p, kFamRowblock, st, [&](double *ys) {
        const int zz = next_reverse(p) ? 64 : 0;
        launch_rowblock_stream(st, p.A, vec, rpb, base_flags | (policy_for(p, kFamRowblock) << 4) | zz, 1.0, trial_beta(), x, ys, dg, cache_ends);
      }))
    return false;
  const int zz = next_reverse(p) ? 64 : 0;
  launch_rowblock_stream(st, p.A, vec, rpb, base_flags | (policy_for(p, kFamRowblock) << 4) | zz, alpha, beta, x, y, dg, cache_ends);
  return true;
}

// adaptive-plus: analysis (row blocks) + the two per-matrix timings.  MIN_NNZ_PER_BLOCK decides how full a block's
// 2048-product tile gets: 1024 (the reference's instance) half-fills it, 1920 fills it but pushes blocks with one longer
// row into a second round; which wins depends on the row-length law (FEM-like: 1536-1920, -10..-18 % time; power-law: 1024),
// so the candidates are timed once per matrix like the cache policy.
bool run_plus_prepare(Plan &p, const int *h_rowptr, hipStream_t st, const double *x) {
  auto launch = [&](int pol, double *ys) {
    launch_plus(st, p.A, p.d_pbp, p.d_pfbr, p.d_pblk, p.plus_blocks, p.plus_has_long, tun(kT_xcd_chunk_tiles), pol,
                p.d_ppartial, 1.0, trial_beta(), x, ys, next_reverse(p));
  };
  const int forced = tun(kT_plus_min_nnz);
  if (forced > 0 || tun(kT_plus_ref_vec)) {
    return ensure_plus(p, h_rowptr, st, forced > 0 ? forced : kPlusMinNnz) && autotune_policy(p, kFamPlus, st, launch);
  }
  if (tun(kT_deterministic)) {
    // by rule, and a pure function of the matrix whatever was called on it before: the balance probe of the row-block shape
    // decides (hub rows: the reference's 1024, the block size that wins on power-law matrices; else 1536)
    int vec = 1, rpb = kThreads;
    pick_rowblock_shape(p.A.m, p.A.nnz, tun(kT_rowblock_target), &vec, &rpb);
    return probe_rowblock(p, rpb, st) && ensure_plus(p, h_rowptr, st, p.rowblock_ok == 0 ? kPlusMinNnz : 1536);
  }
  if (p.plus_tuned_min > 0) return ensure_plus(p, h_rowptr, st, p.plus_tuned_min) && autotune_policy(p, kFamPlus, st, launch);
  // (inside a capture: the row blocks the plan already holds, whatever block size they were analysed with; none yet -> refused)
  if (t_capturing) return (p.plus_blocks >= 0 || ensure_plus(p, h_rowptr, st, 1536)) && autotune_policy(p, kFamPlus, st, launch);
  // (coarse: 1024 where the balance probe found hub rows -- the block size that wins on power-law matrices -- else 1536)
  if (t_coarse_tuning || defer_tuning()) // (budget spent: the coarse choice for now, the three block sizes are timed by a later call)
    return ensure_plus(p, h_rowptr, st, p.rowblock_ok == 0 ? kPlusMinNnz : 1536) && autotune_policy(p, kFamPlus, st, launch);
  // first call on this matrix: cache policy on the middle candidate, then the three block sizes under that policy
  if (!ensure_plus(p, h_rowptr, st, 1536) || !autotune_policy(p, kFamPlus, st, launch)) return false;
  double *scratch = nullptr;
  if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
  TuneTimer timer;
  timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
  bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
  const int candidates[3] = {1536, 1920, kPlusMinNnz};
  float best = 1e30f;
  int best_min = kPlusMinNnz;
  for (int c = 0; ok && c < 3; ++c) {
    ok = ensure_plus(p, h_rowptr, st, candidates[c]);
    if (!ok) break;
    float ms = 0.f;
    ok = timer.time(st, [&] { launch(policy_for(p, kFamPlus), scratch); }, &ms);
    if (ok) tune_log("m %d nnz %d beta class %d row-block-plus: MIN_NNZ_PER_BLOCK %d -> %.2f us (kept for both classes)", p.A.m, p.A.nnz, t_beta_class, candidates[c], ms * 1e3f);
    if (ok && ms < best) {
      best = ms;
      best_min = candidates[c];
    }
  }
  if (!ok) return false;
  p.plus_tuned_min = best_min;
  return ensure_plus(p, h_rowptr, st, best_min);
}

thread_local bool t_in_slab = false; // this thread is running one slab of a column-slab SpMV (no nesting)
bool ensure_segments(Plan &p, int S, hipStream_t st);
// the S passes over the plan's run lists (k_segment.hip); p.seg_state == 1
void run_segments(hipStream_t st, Plan &p, double alpha, double beta, const double *x, double *y) {
  launch_guard_check(st, p.A); // (the passes read run lists, not rowptr: the caller's rowptr is checked here)
  if (beta != 1.0 || p.A.yin) launch_scale_y(st, p.A.m, beta, y, p.A.yin);
  for (int s = 0; s < p.seg_slabs; ++s) {
    if (p.seg_entries[s] == 0) continue;
    launch_segment_tiles(st, p.seg_blocks[s], alpha, p.seg_blk[s], p.seg_row[s], p.seg_begin[s], p.seg_vptr[s], p.A.ci, p.A.v, x, p.d_seg_ys, y);
    if (p.seg_pieces[s]) launch_segment_merge(st, p.seg_entries[s], p.seg_row[s], p.d_seg_ys, y);
  }
}
// automatic mode: slabs of about 32 MB of x (R-MAT scale 25, x = 256 MB, S = 4 / 8 / 12 / 16: 5.92 / 5.31 / 5.59 / 6.08 ms, 7.27 without;
// scale 24, x = 128 MB, S = 4 / 8: 2.37 / 2.59 ms, 3.21 without)
int seg_auto_slabs(int n) {
  const long long s = (static_cast<long long>(n) * 8 + (16LL << 20)) / (32LL << 20);
  return s < 2 ? 2 : (s > 16 ? 16 : static_cast<int>(s));
}

bool run_plus(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y) {
  p.A.cold = nullptr; // (the prepare timings run without hints)
  auto launch_here = [&](double a, double b, double *yy) {
    launch_plus(st, p.A, p.d_pbp, p.d_pfbr, p.d_pblk, p.plus_blocks, p.plus_has_long, tun(kT_xcd_chunk_tiles), policy_for(p, kFamPlus),
                p.d_ppartial, a, b, x, yy, next_reverse(p));
  };
  // Builds the run lists (once) and times the slab passes against this kernel as it stands; ms[0] row-block-plus, ms[1] the passes.
  // Returns false on an error; *timed says whether both timings exist (no room for the lists / rows not ordered: they do not).
  auto time_against_segments = [&](float ms[2], bool *timed) {
    *timed = false;
    // the lists are an optimisation: a matrix that leaves no room for them (or for the build's S x (m + 1) temporaries) keeps the
    // one-kernel path instead of failing the SpMV
    const int S_auto = seg_auto_slabs(p.A.n);
    if (p.seg_state < 0) {
      size_t free_b = 0, total_b = 0;
      const size_t build_bytes = (2 * static_cast<size_t>(S_auto) + 4) * (static_cast<size_t>(p.A.m) + 1) * sizeof(int) + (static_cast<size_t>(p.A.nnz) / 4) * 12;
      const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 2 * build_bytes;
      (void)hipGetLastError();
      if (room && last_error_code_only() == kOk && !ensure_segments(p, S_auto, st)) {
        (void)hipGetLastError();
        tune_log("m %d nnz %d: slab_segments: the run lists could not be built (%s), row-block-plus stays", p.A.m, p.A.nnz, last_error_string());
        clear_error();
        p.free_segments();
      }
    }
    if (p.seg_state != 1) return true;
    ++t_plan_work;
    double *scratch = nullptr;
    if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
    TuneTimer timer;
    timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
    bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
    const double *keep_yin = p.A.yin;
    p.A.yin = nullptr; // (the trial runs update the scratch vector in place)
    ok = ok && timer.time(st, [&] { launch_here(1.0, trial_beta(), scratch); }, &ms[0], /*at_least=*/2) &&
         timer.time(st, [&] { run_segments(st, p, 1.0, trial_beta(), x, scratch); }, &ms[1], /*at_least=*/2);
    p.A.yin = keep_yin;
    *timed = ok;
    return ok;
  };
  const bool slabs_auto = tun(kT_slab_segments) < 0 && !t_in_slab;
  // Power-law columns (the column census finds a hot set, x far beyond the L2s): this kernel is bound by gathers that miss, and the slab
  // passes over run lists (k_segment.hip) usually replace it.  So the passes are decided FIRST, against this kernel in its COARSE
  // configuration -- the rule's cache policy and block size, no hints: nothing timed for it -- and the kernel's own choices (three block
  // sizes, three cache policies, hints: eight trial launches of 7-8 ms each on R-MAT 25, 60 of the 145 ms its first call took) are only
  // timed when the passes do not win clearly.  Clearly = by 15 %: tuned and hinted, this kernel gains up to ~12 % on its coarse form.
  if (slabs_auto && p.seg_choice < 0 && !t_capturing && !by_rule() && tun(kT_gather_hint) != 0) {
    if (!ensure_hint(p, st)) return false;
    if (p.hint_state == 1) {
      const bool was_coarse = t_coarse_tuning;
      t_coarse_tuning = t_no_policy_timing = true;
      const bool prepared = run_plus_prepare(p, h_rowptr, st, x);
      t_coarse_tuning = was_coarse;
      t_no_policy_timing = false;
      if (!prepared) return false;
      float ms[2] = {0.f, 0.f};
      bool timed = false;
      if (!time_against_segments(ms, &timed)) return false;
      if (timed && ms[1] < 0.85f * ms[0]) p.seg_choice = 1;
      if (timed)
        tune_log("m %d nnz %d beta class %d: row-block-plus (coarse) %.2f us, %d column-slab passes over run lists %.2f us -> %s", p.A.m, p.A.nnz, t_beta_class,
                 ms[0] * 1e3f, seg_auto_slabs(p.A.n), ms[1] * 1e3f, p.seg_choice == 1 ? "slab passes" : "not decided: the kernel is tuned first");
      if (!timed) p.seg_choice = 0;
    }
  }
  if (slabs_auto && p.seg_choice == 1) {
    // (a plan that adopted the choice from the tune cache builds its lists here; inside a capture only lists that exist are used)
    if (p.seg_state != 1 && !t_capturing && last_error_code_only() == kOk && !ensure_segments(p, seg_auto_slabs(p.A.n), st)) {
      (void)hipGetLastError(); // (no room for the lists this time: the one-kernel path)
      clear_error();
      p.free_segments();
      p.seg_choice = 0;
    }
    if (p.seg_state == 1) {
      run_segments(st, p, alpha, beta, x, y);
      return true;
    }
  }
  if (!run_plus_prepare(p, h_rowptr, st, x)) return false;
  if (!autotune_hint(p, kFamPlus, st, [&](double *ys) { launch_here(1.0, trial_beta(), ys); })) return false;
  // the passes were not clearly faster than the coarse kernel: once more against the tuned one, the faster (by 5 %) stays
  if (slabs_auto && p.hint_state == 1 && p.seg_choice < 0 && !t_capturing && !by_rule()) {
    // (also while adaptive is timing its kernel families: row-block-plus is then timed as what it will run -- on R-MAT 25 the one-kernel
    // path beats flat by 0.5 % only, 7.26 against 7.29 ms, and a family choice made on that would never meet the 5.3 ms of the passes)
    float ms[2] = {0.f, 0.f};
    bool timed = false;
    if (!time_against_segments(ms, &timed)) return false;
    p.seg_choice = timed && ms[1] < 0.95f * ms[0] ? 1 : 0;
    if (timed)
      tune_log("m %d nnz %d beta class %d: row-block-plus %.2f us, %d column-slab passes over run lists %.2f us -> %s", p.A.m, p.A.nnz,
               t_beta_class, ms[0] * 1e3f, seg_auto_slabs(p.A.n), ms[1] * 1e3f, p.seg_choice ? "slab passes" : "row-block-plus");
    if (p.seg_choice == 0) p.free_segments(); // (the lists of a matrix that does not use them: 12 B per run back)
    if (p.seg_choice == 1 && p.seg_state == 1) {
      run_segments(st, p, alpha, beta, x, y);
      return true;
    }
  }
  launch_here(alpha, beta, y);
  return true;
}

// adaptive, measured: the reference decides from four rowptr samples which kernel family a matrix gets (adaptive.cpp:16-67).
// Which family wins depends on more than those samples say -- fixed row blocks on evenly filled matrices, blocks cut by
// non-zero count where the density varies (quarters, stripes), non-zero-cut tiles where many rows are hundreds to thousands
// long (lognormal row lengths with sigma >= 1: flat 116 us, row blocks 127-130 us; 2000 rows of 3000 nnz in an FEM-like matrix:
// 125 vs 149 us, tools/rowlaw_bench.py) -- so the three are timed once per matrix, each after its own first call has built
// and tuned its plan, and the fastest is kept.  The sample-based rules remain as the untimed form (tunable adaptive_timed 0).
bool run_adaptive_timed(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y) {
  auto run_family = [&](int f, double a, double b, double *yy) {
    switch (f) {
    case 0: return run_rowblock(st, p, h_rowptr, a, b, x, yy);
    case 1: return run_plus(st, p, h_rowptr, a, b, x, yy);
    default: return run_flat(st, p, a, b, x, yy);
    }
  };
  const int cls = t_beta_class;
  if (p.adaptive_family[cls] < 0 && t_capturing) {
    // no timing inside a capture: the family the other beta class settled on (prepared matrices: spmv_acc_prepare times beta = 1),
    // whose plan exists; with neither class timed the call is refused
    if (p.adaptive_family[cls ^ 1] >= 0) return run_family(p.adaptive_family[cls ^ 1], alpha, beta, y);
    return plan_work_allowed("adaptive: timing the kernel families on this matrix");
  }
  // The comparison has two parts: a FIRST LOOK (each family built with its default sub-choices and timed once) and a SECOND LOOK at every
  // family within 8 % of the fastest.  Under the call's tuning budget (defer_tuning) the second look may fall to a later call: the first
  // look's choice serves until then (adaptive_provisional) and its timings are kept in the plan.
  auto decide = [&](const float *ms) {
    // fixed row blocks unless another family is at least 3 % faster (short kernels time within ~2 %)
    int best_family = ms[0] < 1e29f ? 0 : 1;
    for (int f = 1; f < 3; ++f)
      if (ms[f] < (best_family == 0 ? 0.97f * ms[0] : ms[best_family])) best_family = f;
    return best_family;
  };
  // The first look is incremental under the budget: a family that has not been timed yet is built and timed only while the call may still
  // spend (the first call times fixed row blocks at least; the others follow, one per later call if need be), and until all three are in, the
  // best of the measured ones serves.
  float *ms = p.adaptive_ms[cls];
  bool unmeasured = false;
  for (int f = 0; f < 3; ++f) unmeasured = unmeasured || (ms[f] > 1e29f && !p.adaptive_skipped[cls][f]);
  const bool open = p.adaptive_family[cls] < 0 || p.adaptive_provisional[cls];
  if (open && !t_capturing && !(p.adaptive_family[cls] >= 0 && defer_tuning())) {
    ++t_plan_work;
    double *scratch = nullptr;
    if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
    TuneTimer timer;
    timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
    bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
    // The families are compared in the caller's beta class: with beta != 0 every row also reads its old y, which is a large
    // share of the traffic where rows hold one or two non-zeros and ranks the families differently (15 M rows of ~1 nnz:
    // flat looked 3 % faster than the row blocks at beta = 0 and is 9 % slower at beta = 1).
    const double beta_trial = trial_beta();
    t_coarse_tuning = true;
    bool any_measured = false;
    int timed_here = 0;
    for (int f = 0; f < 3; ++f) any_measured = any_measured || ms[f] < 1e29f;
    for (int f = 0; ok && f < 3; ++f) {
      if (ms[f] < 1e29f || p.adaptive_skipped[cls][f]) continue;
      // (at least one family is timed whatever the budget: the call needs a kernel.  A further one is only started while most of the budget is
      // left: building a family's plan is structural work of unknown size -- the row-block analysis, its tables and their allocations took 4 ms on
      // the headline matrix, 25 SpMVs' worth)
      // Every call that gets here times at least ONE family more (progress is guaranteed whatever the budget).
      if (any_measured && timed_here > 0 && (defer_tuning() || budget_spent_fraction() > 0.25)) {
        t_tuning_deferred = true;
        break;
      }
      if (any_measured && timed_here == 0 && p.calls <= 1 && (defer_tuning() || budget_spent_fraction() > 0.25)) { // (the FIRST call: one family is enough)
        t_tuning_deferred = true;
        break;
      }
      ok = run_family(f, 1.0, beta_trial, scratch); // builds this family's plan (sub-choices at their defaults)
      if (!ok) break;
      if (f == 0 && p.rowblock_ok == 0) { // fixed row blocks were rescued: that run WAS family 1
        p.adaptive_skipped[cls][0] = true;
        continue;
      }
      ok = timer.time(st, [&] { (void)run_family(f, 1.0, beta_trial, scratch); }, &ms[f]);
      any_measured = any_measured || ok;
      ++timed_here;
    }
    unmeasured = false;
    for (int f = 0; f < 3; ++f) unmeasured = unmeasured || (ms[f] > 1e29f && !p.adaptive_skipped[cls][f]);
    // second look at every family within 8 % of the fastest: the choice is kept for the life of the plan, and two families
    // 3 % apart changed places from process to process on single timings (the headline matrix ran fixed row blocks in one
    // run and row-block-plus in the next); the smaller of the two timings counts
    bool looked_twice = false;
    if (ok && !unmeasured && (timed_here == 0 || !defer_tuning())) { // (a call that timed no family takes the second look whatever its budget: progress)
      float fastest = ms[0];
      for (int f = 1; f < 3; ++f) fastest = ms[f] < fastest ? ms[f] : fastest;
      for (int f = 0; ok && f < 3; ++f) {
        if (ms[f] > 1.08f * fastest) continue;
        float again = 1e30f;
        ok = timer.time(st, [&] { (void)run_family(f, 1.0, beta_trial, scratch); }, &again);
        if (ok && again < ms[f]) ms[f] = again;
      }
      looked_twice = ok;
    }
    t_coarse_tuning = false;
    const int best_family = decide(ms);
    tune_log("m %d nnz %d adaptive (beta %s 0): fixed row blocks %.2f us, row-block-plus %.2f us, flat %.2f us -> family %d%s", p.A.m, p.A.nnz,
             beta != 0.0 ? "!=" : "==", ms[0] * 1e3f, ms[1] * 1e3f, ms[2] * 1e3f, best_family,
             looked_twice ? "" : (unmeasured ? " (so far: the other families wait for a later call's tuning budget)" : " (first look; the second look waits for a later call's tuning budget)"));
    if (!ok) return false;
    p.adaptive_family[cls] = best_family;
    p.adaptive_provisional[cls] = !looked_twice;
    if (!looked_twice) t_tuning_deferred = true;
  }
  return run_family(p.adaptive_family[cls], alpha, beta, y);
}

} // namespace

namespace {

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
    launch_segment_count(st, A, bounds, S, cnt, beg, flag, rest_below);
    ok = hip_ok(hipMemcpyAsync(&unordered, flag, sizeof(int), hipMemcpyDeviceToHost, st), "read order flag") &&
         hip_ok(hipStreamSynchronize(st), "sync run counts");
  }
  if (ok && unordered) {
    p.seg_state = 0; // some row's columns do not ascend across a slab boundary: its slab parts are not runs
    tune_log("m %d nnz %d: slab_segments: rows are not ordered by column slab, ordinary path", A.m, A.nnz);
  } else if (ok) {
    for (auto *list : {&p.seg_row, &p.seg_begin, &p.seg_vptr, &p.seg_blk}) list->assign(S, nullptr);
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

} // namespace

// The one plan-resident copy of VALUES is the column slabs' (opt-in).  A caller that changes values in place -- which every other
// plan survives -- refreshes it with this: one scatter pass over the matrix, the slabs' structure (and their plans) stay.
int refresh_values(const int *d_rowptr) {
  std::vector<std::shared_ptr<Plan>> todo;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_plans)
      if (std::get<1>(kv.first) == d_rowptr && kv.second->d_slab_rp) todo.push_back(kv.second);
  }
  hipStream_t st = t_stream;
  for (auto &p : todo) {
    std::lock_guard<std::mutex> plan_lock(p->mu);
    if (!p->d_slab_rp) continue;
    launch_slab_scatter(st, p->A, p->slab_width, p->slab_count, p->d_slab_rp, p->d_slab_off, p->d_slab_ci, p->d_slab_v, /*values_only=*/true);
  }
  return static_cast<int>(todo.size());
}

UnboundedTuningScope::UnboundedTuningScope() { ++t_unbounded_tuning; }
UnboundedTuningScope::~UnboundedTuningScope() { --t_unbounded_tuning; }
FlatSegmentSumScope::FlatSegmentSumScope() : prev(t_flat_segment_sum) { t_flat_segment_sum = true; }
FlatSegmentSumScope::~FlatSegmentSumScope() { t_flat_segment_sum = prev; }

void run_spmv(int strategy, int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
              const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx, double *dy,
              const double *dy_in) {
  if (trans != 0) {
    // the reference never reads `trans` (only operation_none is supported, api/spmv.h:13); it computes
    // the non-transposed product.  Same here, but the mismatch is reported out of band.
    set_error(kErrUnsupportedTrans, "only operation_none is supported; computed y = alpha*A*x + beta*y");
  }
  apply_env_tunables();
  if (m <= 0) return;
  if (m > INT_MAX - (1 << 16)) { // row arithmetic in the kernels is int32 with a workgroup's worth of slack, like nnz
    set_error(kErrTooLarge, "row count does not leave room for block arithmetic in int32; shard the matrix");
    return;
  }
  if (!d_rowptr || !dy || (n > 0 && !dx)) {
    set_error(kErrBadArgument, "null rowptr / x / y");
    return;
  }
  if (dy_in == dy) dy_in = nullptr; // in place after all
  if (dy_in && beta != 0.0) {
    // out of place: every row reads y_in[row] and writes y_out[row] exactly once, so the two vectors may be anything but
    // PARTLY overlapping (a shifted view of the same buffer would let one row's store land on another row's unread old value)
    const uintptr_t a = reinterpret_cast<uintptr_t>(dy_in), b = reinterpret_cast<uintptr_t>(dy), bytes = sizeof(double) * static_cast<uintptr_t>(m);
    if (a < b + bytes && b < a + bytes) {
      set_error(kErrBadArgument, "y_in and y_out overlap without being the same vector");
      return;
    }
  }
  hipStream_t st = t_stream;
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    t_capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
  }
  if (strategy < 0 || strategy >= kStrategyCount) {
    set_error(kErrUnknownStrategy, "unknown strategy id");
    return;
  }
  const std::shared_ptr<Plan> p = get_plan(m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value);
  if (!p) return;
  t_last_plan = p;
  // the timing phases of this call share one scratch y (tune_scratch); it goes when the outermost call returns
  struct ScratchScope {
    bool outer;
    ~ScratchScope() {
      if (outer) release_tune_scratch();
    }
  } scratch_scope{!t_in_slab};
  // one call at a time per matrix: plan fields, the per-matrix timings and the carry buffers of flat / row-block-plus belong
  // to the plan (two host threads on DIFFERENT matrices do not meet here; this lock is never held together with g_mu)
  std::lock_guard<std::mutex> plan_lock(p->mu);
  t_beta_class = beta != 0.0 ? 1 : 0;
  // this call's tuning budget (see defer_tuning): the first call on a matrix may spend first_call_budget SpMV-equivalents on trial launches,
  // a later one later_call_budget while something is still open; spmv_acc_prepare, captures (which time nothing) and `deterministic` are outside it
  struct BudgetScope {
    Plan &p;
    bool outer;
    ~BudgetScope() {
      if (!outer) return;
      if (t_first_trial_ms > 0.f) p.trial_ms = t_first_trial_ms;
      p.tuning_open = t_tuning_deferred; // (a call that deferred nothing has settled everything on its path)
      t_budget_spmvs = 0.0;
      t_budget_ms = -1.0;
    }
  } budget_scope{*p, !t_in_slab};
  if (!t_in_slab) {
    t_call_began = std::chrono::steady_clock::now();
    t_tuning_deferred = false;
    t_first_trial_ms = 0.f;
    const int allowance = p->calls == 0 ? tun(kT_first_call_budget) : tun(kT_later_call_budget);
    t_budget_spmvs = (t_unbounded_tuning > 0 || t_capturing || allowance <= 0) ? 0.0 : static_cast<double>(allowance);
    t_budget_ms = (t_budget_spmvs > 0.0 && p->trial_ms > 0.0) ? t_budget_spmvs * p->trial_ms : -1.0;
  }
  // What the reference's harness calls `pre` (its per-call break-point / analysis cost, benchmark_time.cpp:23-43) is paid here
  // by the FIRST call on a matrix: structural passes + per-matrix timings, all of which end in a synchronisation, so the host
  // time from here to the return of that call is the preparation time (the final launch itself is asynchronous).
  // A later call that builds another family's plan (first flat call after adaptive-plus calls, a changed tunable) counts too.
  struct PrepareClock {
    unsigned work0;
    std::chrono::steady_clock::time_point t0;
    ~PrepareClock() {
      t_last_prepare_us =
          t_plan_work != work0 ? std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() : 0.0;
    }
  } prepare_clock{t_plan_work, std::chrono::steady_clock::now()};
  if (p->calls++ == 0) ++t_plan_work; // the plan itself (nnz / guard samples) was just made by get_plan
  (void)hipGetLastError(); // errors of earlier, unrelated HIP calls of this thread are not this call's
  // A plan owns scratch that its kernels write (flat's carries, row-block-plus partials, the slab passes' partial sums, LIGHT's counter):
  // two SpMVs of one matrix in flight on DIFFERENT streams would share it.  The lock above orders the enqueueing, this orders the execution:
  // a call on another stream than the plan's last one waits for that one's work (an event behind its last launch).  Same stream: nothing.
  if (p->launched && p->last_stream != st && !t_capturing && !t_in_slab) {
    if (!p->order_event && hipEventCreateWithFlags(&p->order_event, hipEventDisableTiming) != hipSuccess) p->order_event = nullptr;
    if (p->order_event && hipEventRecord(p->order_event, p->last_stream) == hipSuccess) (void)hipStreamWaitEvent(st, p->order_event, 0);
    (void)hipGetLastError(); // (the other stream may have been destroyed by its owner: its work is then complete)
  }
  p->last_stream = st;
  p->launched = true;
  // where this call's kernels read the old y (kernels.hpp CsrDev::yin); the plan's lock is held until the launches are enqueued
  struct YinScope {
    CsrDev &A;
    ~YinScope() { A.yin = nullptr; }
  } yin_scope{p->A};
  p->A.yin = beta != 0.0 ? dy_in : nullptr;

  if (p->A.nnz == 0) {
    launch_scale_y(st, m, beta, dy, p->A.yin);
    return;
  }
  if (!d_colindex || !d_value) {
    set_error(kErrBadArgument, "null colindex / value with nnz > 0");
    return;
  }
  if (tun(kT_validate) && !validate_plan(*p, st)) return;
  if (tun(kT_guard_full) && !t_in_slab && !launch_full_guard(*p, st)) return; // (a slab is a derived matrix: its parent was checked)

  if (tun(kT_slab_segments) >= 2 && !t_in_slab) {
    // column-slab blocking without a copy: S passes over the plan's run lists (k_segment.hip), whatever the strategy name
    const int S = tun(kT_slab_segments) > 15 ? 15 : tun(kT_slab_segments); // (+ one plane for the short rows of the two-class form: 16 in all)
    if (last_error_code_only() == kOk && !t_capturing && !ensure_segments(*p, S, st)) {
      // (no room for the lists or their S x (m + 1) build temporaries: the passes are an optimisation, the strategy's own kernel runs)
      (void)hipGetLastError();
      tune_log("m %d nnz %d: slab_segments: the run lists could not be built (%s), ordinary path", m, p->A.nnz, last_error_string());
      clear_error();
      p->free_segments();
      p->seg_state = 0; // (not tried again for this plan)
    }
    if (p->seg_state == 1) {
      run_segments(st, *p, alpha, beta, dx, dy);
      if (t_plan_work != prepare_clock.work0 && p->tune_key) tune_store(*p);
      return;
    }
    // (rows not ordered by column slab: the ordinary path below)
  }

  if (tun(kT_col_slabs) >= 2 && !t_in_slab) {
    // opt-in column-slab blocking: S consecutive SpMVs of this strategy on the plan's slabs, the first one applying beta (and
    // reading y_in), the others accumulating into y.  Each slab is an ordinary matrix with a plan of its own.
    const int S = tun(kT_col_slabs) > 16 ? 16 : tun(kT_col_slabs);
    if (!ensure_slabs(*p, S, st)) return;
    launch_guard_check(st, p->A); // (the slabs' kernels check the slabs: the caller's rowptr is checked here)
    // y = beta * y_in first (nothing to do for beta == 1 in place), then every slab: y_s = alpha * A_s x over the slab's non-empty
    // rows (an ordinary SpMV of a smaller matrix, beta = 0) and y[rowid] += y_s
    if (beta != 1.0 || p->A.yin) launch_scale_y(st, m, beta, dy, p->A.yin);
    t_in_slab = true;
    for (int s = 0; s < S && last_error_code_only() == kOk; ++s) {
      const long long o = p->slab_off[s];
      const int ms = p->slab_rows[s];
      if (ms == 0) continue; // an empty slab adds nothing
      run_spmv(strategy, 0, alpha, 0.0, ms, n, static_cast<int>(p->slab_off[s + 1] - o), nullptr, p->slab_crp[s], p->d_slab_ci + o,
               p->d_slab_v + o, dx, p->d_slab_ys, nullptr);
      if (last_error_code_only() == kOk) launch_slab_merge(st, ms, p->slab_rowid[s], p->d_slab_ys, dy);
    }
    t_in_slab = false;
    t_last_plan = p;
    t_beta_class = beta != 0.0 ? 1 : 0;
    if (t_plan_work != prepare_clock.work0 && p->tune_key) tune_store(*p);
    return;
  }

  const long long avg = static_cast<long long>(p->A.nnz) / m;
  // a resident grid for the two persistent-style legacy kernels: CUs x 8 workgroups of 4 waves
  auto resident_blocks = [&]() {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || cus <= 0) cus = 256;
    (void)hipGetLastError();
    return cus * 8;
  };
  if (strategy == kLight && tun(kT_legacy_kernels)) {
    // LightSpMV (hip-light/light_spmv.cpp:16-41): lanes per row from the average row length (its thresholds: vector_row.cpp's
    // table), rows handed out by the plan's counter
    if (!p->d_light_counter) {
      if (!plan_work_allowed("LIGHT's row counter")) return;
      ++t_plan_work;
      if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&p->d_light_counter), 2 * sizeof(unsigned)), "hipMalloc light counter") ||
          !hip_ok(hipMemsetAsync(p->d_light_counter, 0, 2 * sizeof(unsigned), st), "memset light counter") ||
          !hip_ok(hipStreamSynchronize(st), "sync light counter"))
        return;
    }
    launch_light(st, p->A, classic_vec(avg), resident_blocks(), p->d_light_counter, alpha, beta, dx, dy);
    strategy = -1; // handled
  } else if (strategy == kBlockRowOrdinary && tun(kT_legacy_kernels)) {
    launch_block_row(st, p->A, resident_blocks(), alpha, beta, dx, dy); // hip-block-row-ordinary/spmv_hip_acc_imp.cpp:16-75
    strategy = -1;
  }
  switch (strategy) {
  case -1:
    break;
  case kLight:
  case kVectorRow:
  {
    const int forced_w = tun(kT_vector_width);
    const bool tile_form = tun(kT_vector_tile) != 0;
    const int w = (forced_w >= 1 && forced_w <= 64 && (forced_w & (forced_w - 1)) == 0) ? forced_w
                  : tile_form ? tile_vec(avg) : classic_vec(avg);
    if (tun(kT_rowblock_guard) && !probe_rowblock(*p, kThreads / w, st)) return;
    if (tile_form && p->rowblock_ok == 0 && !tun(kT_rescue_flat)) {
      // very uneven rows (hub rows of a power-law matrix): w lanes walking a row of 10^5 non-zeros serialise the kernel (5.8 ms
      // on a 60 000-row power-law matrix that the other families run in 30 us); same rescue as the row-block family
      run_plus(st, *p, h_rowptr, alpha, beta, dx, dy);
    } else if (tile_form && p->rowblock_ok != 0) {
      // the reference's lane width per row (vector_row.cpp:15-27) on the tile machinery
      const double a = static_cast<double>(p->A.nnz) / m;
      auto launch = [&](int pol, double al, double be, double *yy) {
        launch_vector_tile(st, p->A, m, w, w, a, a, tun(kT_vector_target), tun(kT_xcd_chunk), pol, al, be, dx, yy,
                           next_reverse(*p));
      };
      if (!autotune_policy(*p, kFamVector, st, [&](int pol, double *ys) { launch(pol, 1.0, trial_beta(), ys); })) return;
      launch(policy_for(*p, kFamVector), alpha, beta, dy);
    } else {
      launch_vector_row(st, p->A, m, w, 1, alpha, beta, dx, dy, p->rowblock_ok == 0);
    }
    break;
  }
  case kWfRow:
  case kBlockRowOrdinary:
    launch_wave_row(st, p->A, alpha, beta, dx, dy);
    break;
  case kDefault: // the reference's DEFAULT is its one-lane sequential correctness kernel (hip/spmv_hip_acc_imp.cpp:15-35) and
                 // also what its build ships with (config.cmake:15): here the name gets the general-purpose kernel
    run_rowblock(st, *p, h_rowptr, alpha, beta, dx, dy, true); // engine's choice, like adaptive's even-matrix branch
    break;
  case kThreadRow:
    // KERNEL_STRATEGY THREAD_ROW (hip-thread-row/thread_row.cpp:17-48, thread_row_block.hpp): ONE lane sums each row -- the reference's
    // block-level form: the workgroup's non-zeros staged through LDS by coalesced loads, then a thread per row.  Up to 5.8 non-zeros
    // per row that is the row-block kernel's own shape; beyond, where the reference falls back to a 128-block naive loop, the name
    // keeps its meaning here (rows longer than a wavefront are handed to whole waves by tile_row_sum) unless legacy_kernels is 0
    run_rowblock(st, *p, h_rowptr, alpha, beta, dx, dy, false, tun(kT_legacy_kernels) ? 1 : 0);
    break;
  case kLineEnhance:
  case kLine:
    run_rowblock(st, *p, h_rowptr, alpha, beta, dx, dy);
    break;
  case kFlat:
    run_flat(st, *p, alpha, beta, dx, dy);
    break;
  case kAdaptive: {
    if (!fetch_samples(*p, h_rowptr)) return;
    if (tun(kT_adaptive_timed) && !tun(kT_adaptive_split) && !tun(kT_deterministic)) {
      run_adaptive_timed(st, *p, h_rowptr, alpha, beta, dx, dy);
      break;
    }
    // untimed form: the reference's decision tree on four rowptr samples, re-targeted at this library's kernels
    switch (adaptive_branch(m, p->samples)) {
    case 1:
      // The two row halves differ >= 4x in non-zeros.  The reference answers with two lane widths, one per half
      // (vector_row.cpp:30-38; still available as adaptive_vec_row_sparse_spmv / tunable adaptive_split).  Blocks cut by
      // non-zero count with lanes per row chosen per block fit such a matrix better: on a 2 M-row matrix with halves of 40
      // and 5 nnz/row the split took 185 us, row-block-plus 110 us, flat 110 us, fixed row blocks 120 us.
      if (tun(kT_adaptive_split)) {
        const int half_rows = m / 2;
        const long long a0 = half_rows > 0 ? p->samples.half / half_rows : 0;
        const long long a1 = (static_cast<long long>(p->samples.last) - p->samples.half) / (m - half_rows);
        if (tun(kT_vector_tile)) {
          const double f0 = half_rows > 0 ? static_cast<double>(p->samples.half) / half_rows : 0.0;
          const double f1 = (static_cast<double>(p->samples.last) - p->samples.half) / (m - half_rows);
          launch_vector_tile(st, p->A, half_rows, tile_vec(a0), tile_vec(a1), f0, f1, tun(kT_vector_target),
                             tun(kT_xcd_chunk), policy_for(*p, kFamVector), alpha, beta, dx, dy);
        } else {
          launch_vector_row(st, p->A, half_rows, classic_vec(a0), classic_vec(a1), alpha, beta, dx, dy);
        }
      } else {
        run_plus(st, *p, h_rowptr, alpha, beta, dx, dy);
      }
      break;
    default:
      // 2 (adaptive line), 3 (adaptive line-enhance), 4 (adaptive flat), 5 (line-enhance).  The reference sends
      // branch 4 (nnz > 2^23) to flat because its row-block kernels lose balance on large irregular matrices; here
      // the row-block kernel carries a plan-time balance probe and falls back to the row-block-plus kernel exactly
      // then, and measures 1-5 % faster than flat on the balanced large-set stand-ins (one kernel, no carry
      // fix-up), so every non-split branch goes through it.
      // Fixed row blocks are sized from the matrix-wide average row length; where the four row quarters (the samples the
      // decision already holds) differ 1.75x or more in non-zeros, blocks cut by non-zero count fit better: row-block-plus
      // measures 3-7 % faster at 2x-3x (tools/halves_bench.py), the same within 1 % at 1.5x.
      if (quarters_uneven(p->samples) && !tun(kT_adaptive_split)) run_plus(st, *p, h_rowptr, alpha, beta, dx, dy);
      else run_rowblock(st, *p, h_rowptr, alpha, beta, dx, dy, true);
      break;
    }
    break;
  }
  case kAdaptivePlus:
    run_plus(st, *p, h_rowptr, alpha, beta, dx, dy);
    break;
  default:
    set_error(kErrUnknownStrategy, "unknown strategy id");
    break;
  }
  if (t_plan_work != prepare_clock.work0 && p->tune_key) tune_store(*p); // (plan work happened: keep what was learnt)
  // a launch that failed (bad grid, no code object for this device) leaves y untouched: say so
  const hipError_t launch_err = hipGetLastError();
  if (launch_err != hipSuccess && last_error_code_only() == kOk)
    set_error(kErrHip, std::string("kernel launch failed: ") + hipGetErrorString(launch_err));
}

void release_plans(const int *d_rowptr) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto it = g_plans.begin(); it != g_plans.end();) {
    if (!d_rowptr || std::get<1>(it->first) == d_rowptr) {
      it = g_plans.erase(it);
    } else {
      ++it;
    }
  }
  drain_deferred_locked(); // the plans of matrices derived from the ones just dropped
}

bool query_plan(const int *d_rowptr, int m, PlanInfo *out) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto &kv : g_plans) {
    if (std::get<1>(kv.first) == d_rowptr && std::get<4>(kv.first) == m) {
      const Plan &p = *kv.second;
      out->nnz = p.A.nnz;
      out->adaptive_branch = p.have_samples ? adaptive_branch(m, p.samples) : 0;
      int rb_vec = 1, rb_rows = kThreads;
      pick_rowblock_shape(m, p.A.nnz, tun(kT_rowblock_target), &rb_vec, &rb_rows);
      out->vec = rb_vec; // lanes per row of the row-block family for this matrix
      out->flat_tiles = p.flat_tiles;
      out->plus_blocks = p.plus_blocks;
      out->aligned16 = p.A.aligned16 ? 1 : 0;
      // the policy of the family that runs this matrix: adaptive's choice if it was timed, else the first family tuned
      const int afam = p.adaptive_family[1] >= 0 ? p.adaptive_family[1] : p.adaptive_family[0]; // (beta != 0 first: the reference's protocol)
      int fam = afam;
      for (int f = 0; fam < 0 && f < kFamilyCount; ++f)
        if (p.stream_policy[f][1] >= 0 || p.stream_policy[f][0] >= 0) fam = f;
      out->stream_policy = fam < 0 ? -1 : (p.stream_policy[fam][1] >= 0 ? p.stream_policy[fam][1] : p.stream_policy[fam][0]);
      out->flat_fixup = p.flat_tiles > 0 ? (p.flat.needs_fixup ? 1 : 0) : -1;
      out->adaptive_family = afam;
      out->adaptive_family_beta0 = p.adaptive_family[0];
      out->slab_passes = p.seg_state == 1 && (tun(kT_slab_segments) >= 2 || (tun(kT_slab_segments) < 0 && p.seg_choice == 1)) ? p.seg_slabs - (p.seg_rest_below > 0 ? 1 : 0) : 0; // (column slabs: the whole-row pass of the two-class form is not counted)
      return true;
    }
  }
  return false;
}

int cached_plan_count() {
  std::lock_guard<std::mutex> lk(g_mu);
  return static_cast<int>(g_plans.size());
}

} // namespace spmv_acc
