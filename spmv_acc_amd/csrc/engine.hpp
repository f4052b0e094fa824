// engine.hpp -- host-side engine behind both API surfaces (the C ABI of include/spmv_acc.h and the
// C++ mirror of the reference's src/acc/api + per-strategy wrappers).
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>
#include <vector>

#include "kernels.hpp"

namespace spmv_acc {

// The reference's KERNEL_STRATEGY names (src/configure.cmake:17-40) + the benchmark-only
// csr-adaptive-plus entry (benchmark_spmv_acc.hpp:186-200).
enum Strategy {
  kDefault = 0,
  kAdaptive = 1,
  kThreadRow = 2,
  kWfRow = 3,
  kBlockRowOrdinary = 4,
  kLight = 5,
  kVectorRow = 6,
  kLineEnhance = 7,
  kLine = 8,
  kFlat = 9,
  kAdaptivePlus = 10,
  kStrategyCount = 11
};

// error codes reported out-of-band (the reference API returns void and checks nothing)
enum Error {
  kOk = 0,
  kErrUnsupportedTrans = 1,
  kErrBadArgument = 2,
  kErrHip = 3,
  kErrTooLarge = 4,
  kErrUnknownStrategy = 5,
  kErrNoDevice = 6
};

const char *strategy_name(int s);
// Case-insensitive substring match in the reference's order (configure.cmake:18-37 uses regex
// MATCHES, so "line_enhance" is tested before "line"); returns -1 if nothing matches.
int parse_strategy(const char *name);

void set_error(int code, const std::string &what);
int last_error();
const char *last_error_string();
void clear_error();

// measurement switches (see config.cpp g_tunables); -1 for an unknown name
int set_tunable(const char *name, int value);
int get_tunable(const char *name);
void reset_tunables();

// While one of these lives, run_spmv calls of this thread spend whatever the per-matrix timings cost (no first_call_budget / later_call_budget):
// spmv_acc_prepare exists to pay for everything up front.
struct UnboundedTuningScope {
  UnboundedTuningScope();
  ~UnboundedTuningScope();
};

// Host microseconds the calling thread's most recent run_spmv spent preparing its matrix (plan build + per-matrix timings);
// 0 when the plan already existed.
double last_prepare_us();

// the calling host thread's library stream (thread-local, like HIP's current device; NULL until set)
void set_stream(hipStream_t s);
hipStream_t get_stream();

// process-wide strategy: build-time KERNEL_STRATEGY_* macro, overridden by the environment variable
// SPMV_ACC_KERNEL_STRATEGY at first use, overridden by set_active_strategy().
int active_strategy();
int set_active_strategy(int s);

// While one of these lives, kFlat calls of this thread reduce their tiles by the segmented scan (tunable flat_reduce = 1 for the
// thread): how segment_sum_flat_sparse_spmv (reference: hip-flat/flat.cpp:59-76) differs from flat_sparse_spmv.
struct FlatSegmentSumScope {
  FlatSegmentSumScope();
  ~FlatSegmentSumScope();
  bool prev;
};

// Host samples of rowptr that the reference's pickers read from h_csr_desc (adaptive.cpp:24-27,
// flat.cpp:51-52).
struct RowptrSamples {
  int q1 = 0;   // rowptr[m/4]
  int half = 0; // rowptr[m/2]
  int q3 = 0;   // rowptr[3m/4]
  int last = 0; // rowptr[m] = nnz
};

// Which branch adaptive_sparse_spmv takes (adaptive.cpp:30-66): 1 vector-row split, 2 adaptive line,
// 3 adaptive line-enhance, 4 adaptive flat, 5 line-enhance.
int adaptive_branch(int m, const RowptrSamples &s);

// ---- host form of the row-block preprocessing pass ----------------------------------------------------
// Bit-identical to csr_adaptive_plus_analyze_imp (csr_adaptive_plus_analyze.cpp:13-98).
// break_points gets blocks+1 entries, first_block_of_row m+1 entries.  Returns the block count.
int plus_analyze_host(int m, int min_nnz_per_block, int threads_per_block, int vec_size, const int *host_row_ptr,
                      std::vector<int> &break_points, std::vector<int> &first_block_of_row);
// Device form of the same analysis into caller-provided device tables (d_bp: bp_cap entries, d_fbr: m + 1).
// Returns the block count, -1 if bp_cap is too small, -2 on a HIP error.  Synchronises the library stream.
int plus_analyze_device(int m, int min_nnz_per_block, int threads_per_block, int vec_size, const int *d_rowptr,
                        int *d_break_points, int bp_cap, int *d_first_block_of_row);
int plus_pick_vec(int m, int nnz);       // csr_adaptive_plus_spmv.cpp:139-165
int plus_pick_vec_tuned(int m, int nnz, int min_nnz); // row cap chosen so blocks close on MIN_NNZ_PER_BLOCK, not on the cap

// ---- execution ------------------------------------------------------------------------------------------
// One SpMV y = alpha*A*x + beta*y with the given strategy.  h_rowptr may be null: the four samples
// and (for adaptive-plus) the whole rowptr are then fetched from the device once and cached in the
// plan.  nnz < 0 means "unknown": it is read from d_rowptr[m] once.
// dy_in: where the old y is read (out-of-place form y_out = alpha*A*x + beta*y_in); null or == dy: in place, the reference's form.
void run_spmv(int strategy, int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
              const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx, double *dy,
              const double *dy_in = nullptr);

// Drop cached plans (all, or those keyed on this rowptr).  Call when a matrix' structure changes in
// place or its buffers are freed.
void release_plans(const int *d_rowptr, int m_only = -1);
// config.cpp: the kernel clock behind SPMV_ACC_LAUNCH (kernels.hpp), per host thread
void kernel_clock_begin();          // forget the pairs handed out so far (the pool of events is kept)
void kernel_clock_set(bool on);     // on: every launch of this host thread carries its own event pair
void kernel_clock_release();        // destroy the pool's events (end of a timing call)
size_t kernel_clock_used();         // events handed out since kernel_clock_begin (2 per launch)
bool kernel_clock_failed();
hipEvent_t kernel_clock_event(size_t i);

// Introspection for tests / benchmarks.
struct PlanInfo {
  int nnz = 0;
  int adaptive_branch = 0;
  int vec = 0;
  int flat_tiles = 0;
  int plus_blocks = 0;
  int aligned16 = 0;
  int stream_policy = -1; // kStreamPolicy* chosen by the plan-time timing, -1 = not tuned yet
  int flat_fixup = -1;    // 1: flat folds cut rows with the fix-up kernel, 0: tiles finish them, -1: no flat plan yet
  int adaptive_family = -1; // adaptive's timed choice: 0 fixed row blocks, 1 row-block-plus, 2 flat, -1 not timed (the beta != 0
                            // class if it has been timed, else the beta == 0 class)
  int adaptive_family_beta0 = -1; // ... of the beta == 0 class alone
  int slab_passes = 0;            // column slabs whose run lists the plan holds and uses (k_segment.hip), 0 = none
  int last_kernel = -1;           // the kernel behind the plan's latest SpMV (SPMV_ACC_KERNEL_* in include/spmv_acc.h), -1 = none yet
  int col16 = -1;                 // the plan's latest SpMV read the 16-bit column encoding: record ints (16 / 32 / 64), 0 = colindex, -1 = no SpMV yet
  int settled = 0;                // 1: the plan's latest call left no per-matrix timing open (first_call_budget / later_call_budget); 0: later calls will resume some
};
bool query_plan(const int *d_rowptr, int m, PlanInfo *out);
unsigned plan_work_count(); // this thread's count of once-per-matrix steps (structural passes, probes, timing phases) that really ran: unchanged across a call = launches only
int cached_plan_count();
// Persistent per-matrix choices (plan.cpp "tune cache"): path of the text file, null / "" = off; default = environment
// variable SPMV_ACC_TUNE_CACHE.
void set_tune_cache(const char *path);
// Re-copy the caller's VALUES into the plan-resident column slabs of this matrix (tunable col_slabs; no other plan data holds values).
// Enqueued on the calling thread's library stream; returns the number of plans refreshed.
int refresh_values(const int *d_rowptr);
// Drops every cached plan whose stale flag is up (any thread's) and returns how many; records SPMV_ACC_ERR_BAD_ARGUMENT if any.
int check_plans();

} // namespace spmv_acc
