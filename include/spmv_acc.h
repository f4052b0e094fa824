/*
 * spmv_acc.h -- C ABI of the MI355X-native CSR SpMV engine (libspmv_acc.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++ or torch types.  Every entry point
 * names the reference interface (hpcde/spmv-acc, paths relative to the reference root) it replaces.
 * All matrix / vector pointers are DEVICE pointers (hipMalloc) unless a parameter says "host".
 * fp64 values, int32 indices, y = alpha*A*x + beta*y for any alpha, beta (src/acc/api/spmv.h:13-18).
 *
 * Calls are asynchronous: kernels are enqueued on the library stream (NULL stream unless
 * spmv_acc_set_stream was called) and the function returns; the caller synchronises
 * (as cli/main.cpp:104,111 does with hipDeviceSynchronize).  The reference API returns void and
 * checks nothing; errors here are reported out of band through spmv_acc_last_error().
 */
#ifndef SPMV_ACC_C_ABI_H
#define SPMV_ACC_C_ABI_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- KERNEL_STRATEGY plugin surface ------------------------------------------------------------
 * replaces: the compile-time switch -DKERNEL_STRATEGY=<name> (config.cmake:15,
 * src/configure.cmake:17-40, src/building_config.h.in:24-34, src/acc/strategy_picker.cpp:19-65).
 * The build-time macro KERNEL_STRATEGY_<NAME> still selects the default; the environment variable
 * SPMV_ACC_KERNEL_STRATEGY and spmv_acc_set_strategy() override it at run time so one binary serves
 * every configuration.  Names are matched like the reference's CMake regex: case-insensitive
 * substring, "line_enhance" tested before "line". */
enum spmv_acc_strategy {
  SPMV_ACC_DEFAULT = 0,
  SPMV_ACC_ADAPTIVE = 1,
  SPMV_ACC_THREAD_ROW = 2,
  SPMV_ACC_WF_ROW = 3,
  SPMV_ACC_BLOCK_ROW_ORDINARY = 4,
  SPMV_ACC_LIGHT = 5,
  SPMV_ACC_VECTOR_ROW = 6,
  SPMV_ACC_LINE_ENHANCE = 7,
  SPMV_ACC_LINE = 8,
  SPMV_ACC_FLAT = 9,
  SPMV_ACC_ADAPTIVE_PLUS = 10 /* benchmark-only entry of the reference (benchmark_spmv_acc.hpp:186-200) */
};
int spmv_acc_set_strategy(const char *name); /* 0 on success, -1 unknown name */
int spmv_acc_set_strategy_id(int strategy);
int spmv_acc_get_strategy(void);
const char *spmv_acc_strategy_name(int strategy);
int spmv_acc_parse_strategy(const char *name); /* -1 if no match */

/* ---- primary entry -------------------------------------------------------------------------------
 * replaces: void sparse_spmv(int htrans, const double halpha, const double hbeta, int hm, int hn,
 *           const int *rowptr, const int *colindex, const double *value, const double *x, double *y)
 *           -- src/acc/api/spmv.h:27-28, src/acc/api/spmv_imp.cpp:10-18 (C++ linkage there; the same
 *           ten arguments with C linkage here).
 * The reference reads rowptr[hm] on the HOST (spmv_imp.cpp:14), which needs host-visible device
 * memory; here nnz and the strategy pickers' rowptr samples are fetched from the device once per
 * matrix and cached, so plain hipMalloc memory works.
 * (A C++ translation unit that has already included api/spmv.h sees that header's C++-linkage declaration;
 * the library exports both symbols, one name cannot carry two linkages in one TU.) */
#ifndef SPMV_ACC_AMD_API_SPMV_H
void sparse_spmv(int htrans, const double halpha, const double hbeta, int hm, int hn, const int *rowptr,
                 const int *colindex, const double *value, const double *x, double *y);
#endif

/* ---- descriptor entry, flattened to C --------------------------------------------------------------
 * replaces: void sparse_csr_spmv(int trans, const double alpha, const double beta,
 *           const csr_desc<int,double> h_csr_desc, const csr_desc<int,double> d_csr_desc,
 *           const double *dx, double *dy) -- src/acc/api/spmv.h:20-21, strategy_picker.cpp:19-65
 *           (the entry spmv-cli calls, cli/main.cpp:102,110,117).
 * h_rowptr: HOST copy of rowptr or NULL.  nnz: number of non-zeros, or -1 to read rowptr[m]. */
void spmv_acc_csr_spmv(int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
                       const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx,
                       double *dy);

/* ---- per-strategy entry ------------------------------------------------------------------------------
 * replaces: the L1 wrappers the benchmark harness calls directly (benchmark/benchmark_spmv_acc.hpp:27-200):
 *   default_sparse_spmv (hip/spmv_hip_acc_imp.cpp:29), adaptive_sparse_spmv (hip-adaptive/adaptive.cpp:16),
 *   flat_sparse_spmv (hip-flat/flat.cpp:47), line_enhance_sparse_spmv / adaptive_enhance_sparse_spmv
 *   (hip-line-enhance/line_enhance_spmv.cpp:8,23), adaptive_line_sparse_spmv (hip-line/line_strategy.cpp:52),
 *   vec_row_sparse_spmv (hip-vector-row/vector_row.cpp:9), csr_adaptive_plus_sparse_spmv
 *   (hip-csr-adaptive-plus/csr_adaptive_plus_spmv.cpp:132), ... selected by `strategy`. */
void spmv_acc_csr_spmv_strategy(int strategy, int trans, double alpha, double beta, int m, int n, int nnz,
                                const int *h_rowptr, const int *d_rowptr, const int *d_colindex,
                                const double *d_value, const double *dx, double *dy);

/* ---- out-of-place form (new) ---------------------------------------------------------------------------
 *     y_out = alpha * A * x + beta * y_in
 * replaces: nothing callable in the reference -- every entry there updates y in place (api/spmv.h:13-18), and its drivers
 * re-upload y0 before each call (cli/main.cpp:101,116).  A caller that keeps both vectors (x_{k+1} = f(y_k) iterations, the
 * row-sharded step below: old slice in one buffer, new slice inside the gathered vector) would otherwise copy y once per SpMV
 * (2 x 8 B per row; 61 of 218 us per sharded step on the headline matrix).  Same kernels, same sums: bit-identical to the
 * in-place entry on a copy of y_in.  strategy < 0: the active strategy.  dy_in may be NULL or equal to dy_out (in place);
 * the two vectors must not overlap partially (SPMV_ACC_ERR_BAD_ARGUMENT); dy_in is not read when beta == 0. */
void spmv_acc_csr_spmv_oop(int strategy, int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
                           const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx,
                           const double *dy_in, double *dy_out);

/* ---- row sub-ranges of one matrix as consecutive launches over two streams (new) ------------------------------------------
 * replaces: nothing in the reference (one kernel per SpMV on the NULL stream).  The compute side of the pipelined row-sharded step
 * (spmv_acc_shard_step with pipeline > 1, spmv_acc_amd/dist.py): rows [row_cuts[k], row_cuts[k + 1]) of the matrix are chunk k,
 * handed to the kernels as an un-rebased row sub-range (d_rowptr + row_cuts[k], the whole colindex / value arrays; nnz_ends[k] =
 * rowptr[row_cuts[k + 1]], which the caller reads once); chunk k's kernels go to streams[k & 1] -- consecutive chunks are
 * independent, on one stream each would wait for its predecessor's last wavefront -- and events[k] (a hipEvent_t the caller
 * made; may be NULL) is recorded behind them, so that the caller can send chunk k on its way while chunk k + 1 computes.  ONE
 * host call instead of one per chunk (each costs microseconds of a step that lasts tens).  y_out / y_in as spmv_acc_csr_spmv_oop,
 * indexed by the matrix' rows.  The calling thread's library stream is left as it was.  Returns 0 or the first error. */
int spmv_acc_csr_spmv_chunks(int strategy, double alpha, double beta, int n, int nchunks, const int *row_cuts, const int *nnz_ends,
                             const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx,
                             const double *dy_in, double *dy_out, void *const *streams, void *const *events);

/* ---- row-block preprocessing pass, device form ---------------------------------------------------------
 * replaces: pre_calc_break_point<STRIDE, BLOCKS, int><<<1024,512>>>(row_ptr, m, break_points, bp_len)
 *           -- src/acc/hip-flat/flat_imp.inl:108-131, launched from flat.cpp:25,43.
 * d_break_points (device, bp_len ints) receives bit-identical values; no pre-zeroing needed.
 * Returns 0, or an error code. */
int spmv_acc_break_points(const int *d_rowptr, int m, int nnz, int stride, int *d_break_points, int bp_len);
int spmv_acc_break_points_len(int nnz, int stride); /* flat.cpp:35-38: ceil(nnz/stride) + 1 */

/* ---- row-block preprocessing pass, host form --------------------------------------------------------------
 * replaces: csr_adaptive_plus_analyze_imp<int, THREADS, VEC>(m, nnz, MIN_NNZ_PER_BLOCK, break_points,
 *           first_block_of_row, host_row_ptr, dev_row_ptr) -- hip-csr-adaptive-plus/csr_adaptive_plus_analyze.cpp:13-98.
 * h_break_points: host, capacity bp_cap (m + 2 + nnz / (2 * min_nnz_per_block) is always enough); h_first_block_of_row: host, m + 1 ints.
 * Returns the number of row blocks, -1 if bp_cap is too small. */
int spmv_acc_adaptive_plus_analyze(int m, int min_nnz_per_block, int threads_per_block, int vec_size,
                                   const int *h_rowptr, int *h_break_points, int bp_cap,
                                   int *h_first_block_of_row);
/* The same analysis on the DEVICE (new): next-block search per row + pointer jumping + scan; bit-identical tables,
 * no host rowptr and no PCIe traffic.  d_break_points: device, bp_cap ints; d_first_block_of_row: device, m + 1 ints.
 * Returns the number of row blocks, -1 if bp_cap is too small, -2 on error.  Synchronises the library stream. */
int spmv_acc_adaptive_plus_analyze_device(int m, int min_nnz_per_block, int threads_per_block, int vec_size,
                                          const int *d_rowptr, int *d_break_points, int bp_cap,
                                          int *d_first_block_of_row);
int spmv_acc_adaptive_plus_vec(int m, int nnz); /* csr_adaptive_plus_spmv.cpp:139-165 */

/* ---- strategy pickers (host logic, no GPU needed) --------------------------------------------------------------
 * replaces: the decision tree of adaptive_sparse_spmv, hip-adaptive/adaptive.cpp:24-66, on the same four
 * inputs rowptr[m/4], rowptr[m/2], rowptr[3m/4], rowptr[m].  Returns 1 vector-row split, 2 adaptive line,
 * 3 adaptive line-enhance, 4 adaptive flat, 5 line-enhance. */
int spmv_acc_adaptive_branch(int m, int rp_quarter, int rp_half, int rp_three_quarter, int rp_last);

/* ---- multi-GPU row-range partition (new; the reference is single-GPU) -----------------------------------------------
 * Contiguous row ranges for `parts` ranks.  mode 0: equal row counts (what an allgather of equal-sized
 * y shards needs); mode 1: nnz-balanced boundaries found by binary search on rowptr.
 * h_rowptr: host rowptr (may be NULL for mode 0).  row_begin: out, parts + 1 entries. */
int spmv_acc_partition_rows(int m, int parts, int mode, const int *h_rowptr, int *row_begin);

/* ---- one rank's step of the row-sharded SpMV, behind the C boundary (new) ---------------------------------------------------
 * replaces: nothing in the reference (single GPU: hipSetDevice(0) at cli/main.cpp:89, no collective anywhere); BASELINE's
 * north_star adds the row-range partition with an RCCL allgather of the y sub-vectors.  spmv_acc_amd/dist.py does this with
 * torch.distributed; this entry gives C / C++ consumers (one process or one thread per GPU) the same step:
 *     y_local[0 .. m_local) = alpha * A_local * x + beta * y_local          (this rank's rows, any strategy)
 *     ncclAllGather(y_local, y_full, m_pad doubles, comm, library stream)   (ONE collective per SpMV, behind the kernels)
 * nccl_comm: the caller's ncclComm_t.  The library resolves ncclAllGather at run time from the RCCL the process already has
 * (dlopen RTLD_NOLOAD of librccl.so.1 / librccl.so, else a fresh dlopen; environment variable SPMV_ACC_RCCL_LIB names a
 * particular file), so libspmv_acc.so keeps linking only the HIP runtime.  y_local holds m_pad >= m_local doubles (every rank
 * the same m_pad: RCCL has no allgatherv; rows past m_local are padding the caller zeroes once), y_full holds
 * world * m_pad doubles; rank r's rows land at y_full + r * m_pad.  Returns 0 or an error code (SPMV_ACC_ERR_NO_DEVICE when
 * no RCCL can be found, SPMV_ACC_ERR_HIP when the collective fails). */
int spmv_acc_sharded_spmv(void *nccl_comm, int strategy, double alpha, double beta, int m_local, int m_pad, int n, int nnz_local,
                          const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx,
                          double *dy_local, double *dy_full);

/* ---- the same step with a per-rank handle: in place, out of place, pipelined (new) ---------------------------------------------
 * replaces: nothing in the reference (single GPU).  One handle per rank and matrix, made by the host thread that drives the GPU
 * (hipSetDevice first; the library stream is per host thread).  A step
 *     y_full[rank * m_pad + i] = alpha * (A_local * x)[i] + beta * y_in_local[i]      i in [0, m_local)
 * computes this rank's rows straight into their place in the gathered vector -- dy_in_local (m_local doubles; NULL = that place
 * itself, i.e. in place) is read by the out-of-place kernels, so no slice is ever copied -- and then every rank receives every
 * slice, in place:
 *   pipeline <= 1: ONE ncclAllGather (send buffer = this rank's slice of dy_full) on the library stream, behind the kernels;
 *   pipeline  = C: the local rows are cut into C chunks -- row sub-ranges of the caller's own arrays, nothing is copied or rebased; their
 *                  kernels alternate over two streams of the shard's own (consecutive chunks are independent);
 *                  chunk c's slice travels -- grouped ncclSend / ncclRecv with every peer, straight to its place in their
 *                  vectors, on a second stream -- as soon as its kernels have finished, while chunk c+1 computes.  This is the
 *                  overlap that survives when the next x depends on the gathered y.  The library stream waits for the last
 *                  arrival, so work enqueued after the step sees the whole vector.
 * Rows [m_local, m_pad) of a slice are padding: zero dy_full once.  dy_full holds world * m_pad doubles (world and rank are
 * the communicator's).  RCCL is resolved at run time as in spmv_acc_sharded_spmv.  Returns 0 or an error code.
 * spmv_acc_rccl_comm_init_all / _destroy: ncclCommInitAll / ncclCommDestroy through the same run-time binding, for a
 * one-process driver that must not link RCCL either (spmv-cli --gpus N); devices may be NULL (0 .. ndev-1). */
typedef struct spmv_acc_shard *spmv_acc_shard_t;
int spmv_acc_shard_create(spmv_acc_shard_t *out, void *nccl_comm, int strategy, int m_local, int m_pad, int n, int nnz_local,
                          const int *d_rowptr, const int *d_colindex, const double *d_value, int pipeline);
int spmv_acc_shard_step(spmv_acc_shard_t shard, double alpha, double beta, const double *dx, const double *dy_in_local,
                        double *dy_full);
/* Builds and tunes every chunk's plan for the beta class of `beta` (beta == 0 / beta != 0) with x = dx, so that the steps only enqueue: plan
 * building allocates, frees and synchronises, which must not fall between the exchanges of a step the peers are already in.  No collective
 * inside; every rank calls it once before its first step (spmv-cli --gpus N does).  A rank whose local SpMV fails inside a step still takes
 * part in all of that step's exchanges and reports its error afterwards: the peers are never left waiting in a collective. */
int spmv_acc_shard_prepare(spmv_acc_shard_t shard, double beta, const double *dx);
int spmv_acc_shard_pipeline(spmv_acc_shard_t shard); /* chunks per step actually in use */
int spmv_acc_shard_destroy(spmv_acc_shard_t shard);
int spmv_acc_rccl_comm_init_all(void **comms, int ndev, const int *devices);
int spmv_acc_rccl_comm_destroy(void *comm);

/* ---- host staging (new; replaces the pageable blocking hipMemcpy of cli/utils.hpp:94-117) ----------------------------
 * Copies host CSR arrays + vectors to freshly hipMalloc'ed device buffers: the caller's arrays are pinned in place
 * (hipHostRegister) and sent with hipMemcpyAsync on a private copy stream, all transfers in flight together; an
 * array that cannot be pinned goes through a pinned double buffer.  Any of the host pointers
 * may be NULL to skip that array.  Free with spmv_acc_free_device. */
int spmv_acc_stage_csr(int m, int n, int nnz, const int *h_rowptr, const int *h_colindex, const double *h_value,
                       const double *h_x, const double *h_y, int **d_rowptr, int **d_colindex, double **d_value,
                       double **d_x, double **d_y);
int spmv_acc_free_device(void *p);

/* ---- explicit preprocessing (new) ---------------------------------------------------------------------------------------
 * Builds everything the FIRST call on a matrix would build for `strategy` -- the structural passes (break points, row-block
 * analysis, balance probe) and the per-matrix timings (cache policy, flat's cut-row form, adaptive-plus block size) -- by
 * running that first call into a zeroed scratch y (alpha = beta = 1, the reference's protocol), then synchronises.  The caller's y is not touched; x is read.
 * After it every spmv call on the matrix is kernel launches only (capturable into a hipGraph).  ms_out (may be NULL):
 * device time of the preparation, the figure the reference's benchmark reports as `pre` (benchmark_time.cpp:23-43;
 * there it is the per-call break-point / analysis cost, here it is paid once).
 * replaces: nothing callable in the reference (its preprocessing is re-done inside every SpMV call, flat.cpp:35-45,
 * csr_adaptive_plus_spmv.cpp:104-125). */
int spmv_acc_prepare(int strategy, int m, int n, int nnz, const int *h_rowptr, const int *d_rowptr, const int *d_colindex,
                     const double *d_value, const double *dx, float *ms_out);
/* The same for the beta class the caller will run in: the choices that depend on whether y is read (stream cache policy,
 * adaptive's kernel family, flat's cut-row form) are timed and kept PER CLASS (beta == 0: y only written; beta != 0: read too).
 * spmv_acc_prepare is the beta != 0 class (the reference's protocol, alpha = beta = 1).  A class that was never prepared is
 * timed by its first call -- or, inside a stream capture, runs with the other class' choices. */
int spmv_acc_prepare_beta(int strategy, double beta, int m, int n, int nnz, const int *h_rowptr, const int *d_rowptr,
                          const int *d_colindex, const double *d_value, const double *dx, float *ms_out);

/* ---- persistent choices + the deterministic switch (new) -------------------------------------------------------------------
 * The reference's strategy choice is a pure function of its inputs (strategy_picker.cpp:19-65, adaptive.cpp:24-66) and costs
 * nothing; this library times a handful of choices on each matrix' first call (cache policy, kernel family, tile geometry: up
 * to 13 ms on the headline matrix), per process.  Two opt-in ways out:
 *   spmv_acc_set_tune_cache(path) / environment SPMV_ACC_TUNE_CACHE=<file>: the choices are appended to a text file, one line per
 *     matrix, keyed by a digest of (library version, device name, m, n, nnz, 64 rowptr samples); a later process that meets the
 *     same matrix on the same device adopts them and only runs the structural passes (NULL or "" switches it off);
 *   tunable "deterministic" = 1 / environment SPMV_ACC_DETERMINISTIC=1: nothing is timed at all, every choice follows a fixed
 *     rule on the matrix' shape -- y is then bitwise equal across processes and runs (the kernels never use atomics).
 * Round 6: with the default ("deterministic" = 0) the calls made BEFORE a plan is settled are answered by that same rule (the plan's rule twin)
 * while the timings advance beside them against a scratch y; from the first settled call on the timed choices serve.  y changes its last bits at
 * most once per (matrix, strategy, beta class), at a call spmv_acc_query_plan_settled shows.  "deterministic" = -1: as rounds 2-5 (the timed
 * choices as far as they have come serve from the first call on). */
void spmv_acc_set_tune_cache(const char *path);

/* ---- values changed in place, with the opt-in column slabs in use (new) ---------------------------------------------------------
 * Every plan survives in-place edits of `value` (the reference keeps nothing between calls) -- except the one opt-in mode whose plan
 * holds a re-ordered COPY of the matrix (tunable col_slabs).  After changing values in place, call this instead of dropping the plan:
 * one scatter pass re-copies the values into the slabs (enqueued on the calling thread's library stream, ordered before later
 * SpMVs on it); the slabs' structure, plans and timed choices stay.  Returns the number of plans refreshed (0: the matrix has no
 * slabs, nothing to do).  A changed STRUCTURE (rowptr / colindex) still needs spmv_acc_release_plans. */
int spmv_acc_refresh_values(const int *d_rowptr);

/* Host microseconds the calling thread's most recent SpMV call spent preparing its matrix (structural passes + per-matrix
 * timings of the FIRST call on a matrix); 0 when the plan already existed.
 * replaces: BenchmarkTime::pre of the reference's harness (benchmark/utils/benchmark_time.cpp:23-43,
 * benchmark/flat/spmv_acc_flat.cpp:20-71: the break-point pass timed on EVERY call there) and
 * SpMVAccHanele::profile_analyze_time (csr_adaptive_plus_spmv.cpp:98-128). */
double spmv_acc_last_prepare_us(void);

/* ---- plan cache, stream, errors ------------------------------------------------------------------------------------------
 * Preprocessing results (break points, row blocks, carries) are cached per matrix, keyed by
 * (device, rowptr, colindex, value, m, n).  Release when a matrix' structure changes in place or its
 * buffers are freed; NULL releases everything.
 * Stale-plan guard: the reference recomputes its preprocessing on every call (hip-flat/flat.cpp:39-44), so its callers
 * never announce a change.  Every plan therefore records 64 strided rowptr entries (rowptr[0] .. rowptr[m] = nnz);
 * the first wavefront of every SpMV kernel re-reads them and raises a sticky flag (pinned host memory, no
 * synchronisation) when the matrix behind the pointers is no longer the one the plan was built for -- buffers freed
 * and re-allocated at the same addresses for another matrix of the same shape, or a structure rewritten in place.
 * spmv_acc_last_error() (after the caller's synchronisation) and the next call on those pointers then report
 * SPMV_ACC_ERR_BAD_ARGUMENT ("... changed ..."): the y of the call that ran on the stale plan is invalid, the plan
 * is dropped and the next call builds a fresh one before it runs.  Values edited in place never trip the guard
 * (plans hold no copy of colindex or values).  A change that leaves all 64 samples untouched is not seen: callers
 * that permute a few rows in place still have to call spmv_acc_release_plans. */
void spmv_acc_release_plans(const int *d_rowptr);
int spmv_acc_cached_plans(void);
/* spmv_acc_last_error() asks the plan the CALLING THREAD used last (one load, no lock).  This one looks at every cached plan,
 * any thread's: drops the stale ones, returns how many there were (and records SPMV_ACC_ERR_BAD_ARGUMENT if any).  Call after a
 * device synchronisation. */
int spmv_acc_check_plans(void);
/* plan introspection: fills out[9] = {nnz, adaptive_branch, vec, flat_tiles, plus_blocks, aligned16, stream_policy,
 * flat_fixup, adaptive_family};
 * stream_policy: cache policy of the stream loads chosen by timing at plan time (0 nt, 1 default, 3 values default,
 * -1 not tuned yet); flat_fixup: 1 the flat plan folds cut rows with its fix-up kernel, 0 tiles finish them, -1 no flat
 * plan yet; adaptive_family: the kernel family adaptive settled on by timing (0 fixed row blocks, 1 row-block-plus, 2 flat,
 * -1 not timed); returns 1 if a plan exists */
int spmv_acc_query_plan(const int *d_rowptr, int m, int *out);
/* adaptive's timed kernel family for the beta == 0 class alone (out[8] above reports the beta != 0 class when it has been
 * timed): the families are timed per class because the ranking changes with the y read; -2 = no such plan */
int spmv_acc_query_plan_beta0(const int *d_rowptr, int m);
/* number of column slabs whose run lists the plan holds and its SpMVs pass over (tunable slab_segments, k_segment.hip: column-slab
 * blocking without a copy of the matrix -- chosen by a plan-time timing on matrices whose column census finds a hot set, or forced);
 * 0 = the plan runs the named strategy's own kernel; -2 = no such plan.  No reference counterpart: the reference has one path per
 * strategy (strategy_picker.cpp:19-65). */
int spmv_acc_query_plan_slab_passes(const int *d_rowptr, int m);
/* 1: the latest call on this plan left none of its per-matrix timings open -- the plan is settled, calls are launches only and bitwise stable;
 * 0: the first-call budget (tunable first_call_budget) deferred some, the next calls resume them (or call spmv_acc_prepare); that call was answered
 * by the plan's rule twin (round 6), as the following ones are until one returns 1 here; -2 = no such plan.
 * No reference counterpart (the reference times nothing). */
int spmv_acc_query_plan_settled(const int *d_rowptr, int m);
/* Did the plan's LATEST SpMV read the plan's 16-bit column encoding instead of colindex (round 6; tunable col16, default: timed per matrix and
 * kernel family)?  16 / 32 / 64 = yes, with that many ints per 256-non-zero chunk record; 0 = no (the caller's colindex was streamed);
 * -1 = no SpMV yet; -2 = no such plan.  A plan that uses the encoding holds structure derived from colindex: after editing column indices in
 * place (same rowptr) call spmv_acc_release_plans (64 samples of colindex are re-checked by every launch, like rowptr's).  No reference counterpart:
 * the reference streams one 4-byte column per non-zero (hip-flat/flat_imp_one_pass.hpp:35-39, hip-line-enhance/line_enhance_spmv_imp.inl:55-62). */
int spmv_acc_query_plan_col16(const int *d_rowptr, int m);
/* Which kernel ran the plan's LATEST SpMV (round 5).  The reference's strategy name IS its kernel (strategy_picker.cpp:19-65); here a name selects a
 * policy by default (`flat` may run the row-block kernel where it timed faster, `line_enhance` the column-slab passes on power-law columns) and
 * tunable strict_strategy = 1 (SPMV_ACC_TUNABLES=strict_strategy=1) binds the name to its algorithm.  -1 = no SpMV yet, -2 = no such plan. */
enum spmv_acc_kernel {
  SPMV_ACC_KERNEL_ROWBLOCK = 0,    /* rowblock_stream_kernel: line_enhance / line / thread_row / default (line_enhance_spmv_imp.inl:12-95) */
  SPMV_ACC_KERNEL_ROWBLOCK_PLUS = 1, /* plus_kernel: adaptive_plus, and the row-block family's rescue of unbalanced rows (csr_adaptive_plus_spmv_imp.inl:31-205) */
  SPMV_ACC_KERNEL_FLAT_TILE = 2,   /* flat_tile_kernel (+ fix-up): flat (flat_imp_one_pass.hpp:16-77) */
  SPMV_ACC_KERNEL_SLAB_PASSES = 3, /* segment_tile_kernel passes over the plan's run lists (no reference counterpart) */
  SPMV_ACC_KERNEL_VECTOR_TILE = 4, SPMV_ACC_KERNEL_VECTOR_ROW = 5, SPMV_ACC_KERNEL_WAVE_ROW = 6, SPMV_ACC_KERNEL_LIGHT = 7,
  SPMV_ACC_KERNEL_BLOCK_ROW = 8, SPMV_ACC_KERNEL_COL_SLABS = 9, SPMV_ACC_KERNEL_SCALE_ONLY = 10
};
int spmv_acc_query_plan_last_kernel(const int *d_rowptr, int m);

void spmv_acc_set_stream(void *hip_stream); /* hipStream_t; NULL = the NULL stream (reference behaviour).  The stream belongs to the
                                             * CALLING HOST THREAD, like HIP's current device: N threads driving N GPUs each set
                                             * their own and cannot redirect one another.  Steady-state calls are
                                             * launches only and may be captured into a hipGraph; the first call on a matrix
                                             * (plan: allocations, synchronisation, timings) must run outside a capture: a call that
                                             * would need such work inside a capture enqueues nothing and reports
                                             * SPMV_ACC_ERR_BAD_ARGUMENT (timed choices that are merely missing are skipped instead).
                                             * A thread that never set a stream launches on the NULL stream; if another thread HAS set
                                             * one, the first such launch prints a one-time note on stderr (SPMV_ACC_QUIET=1: none).
                                             * A plan remembers the stream of its latest call to order a call on another stream behind
                                             * it: before destroying a stream, synchronise it or release the plans used on it.
                                             * Captured graphs bypass that ordering: a plan owns scratch its kernels write (flat's
                                             * carries, the slab passes' partial sums, LIGHT's row counter), so two graphs that hold
                                             * SpMVs of ONE matrix must not be replayed concurrently on two streams.
                                             * The first call on a matrix spends at most ~20 SpMV-equivalents on per-matrix timings
                                             * (tunable first_call_budget) and the following calls finish them; until they have, two
                                             * calls may run different kernels, i.e. sum in a different order -- spmv_acc_prepare
                                             * settles everything up front, tunable deterministic times nothing at all. */
void *spmv_acc_get_stream(void);

int spmv_acc_last_error(void); /* 0 = ok; see enum below.  Per host thread, like errno. */
const char *spmv_acc_last_error_string(void);
void spmv_acc_clear_error(void);
enum spmv_acc_error {
  SPMV_ACC_OK = 0,
  SPMV_ACC_ERR_UNSUPPORTED_TRANS = 1,
  SPMV_ACC_ERR_BAD_ARGUMENT = 2,
  SPMV_ACC_ERR_HIP = 3,
  SPMV_ACC_ERR_TOO_LARGE = 4,
  SPMV_ACC_ERR_UNKNOWN_STRATEGY = 5,
  SPMV_ACC_ERR_NO_DEVICE = 6
};

/* ---- measurement helper ----------------------------------------------------------------------------------------------------
 * replaces: hip::timer::event_timer around the L1 call (benchmark/utils/timer_utils.h:16-51,
 * benchmark/csr_spmv.hpp:67-74).  Runs `iters` SpMVs with `strategy`; each is bracketed by hipEvents on
 * the library stream, y is restored from d_y0 (device, m doubles, may be NULL) outside the timed region -- by a non-temporal copy kernel
 * (round 5: like the DMA write of the reference's hipMemcpy reset, csr_spmv.hpp:68, it parks nothing in the L2s; env SPMV_ACC_RESET_NT=0: the
 * default-policy copy of rounds 1-4, SPMV_ACC_RESET_MEMCPY=1: hipMemcpyAsync device -> device).
 * ms_out receives iters per-launch durations in milliseconds.  Returns 0 or an error code.  What is timed is a SETTLED plan: the helpers
 * first finish whatever per-matrix timings the first-call budget left open (spmv_acc_prepare_beta: at least one untimed SpMV into a scratch y). */
int spmv_acc_time_spmv(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                       const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                       const double *dx, double *dy, const double *d_y0, float *ms_out);

/* The same with the events' creation flags chosen by the caller (hipEventCreateWithFlags): 0 = hipEventDefault, what
 * benchmark/utils/timer_utils.h:16-51 creates and what every gate of this repository is quoted on.  hipEventDisableSystemFence
 * (0x20000000) takes the system-scope fence -- a cache write-back and invalidation -- out of the event, which HIP documents as the
 * more accurate form for events that only measure time; on MI355X it is 1.2 us of the ~6.7 us an almost empty launch takes between
 * default events, and it leaves the L2s warm for the timed launch.  Reported beside the default figure, never instead of it. */
int spmv_acc_time_spmv_events(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                              const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                              const double *dx, double *dy, const double *d_y0, float *ms_out, unsigned event_flags);

/* The per-launch protocol with a COLD cache hierarchy (round 6; context for every fraction, never a gate): after y has been restored and before the
 * start event, flush_bytes of scratch traffic under the default cache policy (the copy kernel: second half of d_flush overwritten with the first)
 * displace what the previous launch left in the L2s and the 256 MB Infinity Cache.  The reference's protocol (benchmark/csr_spmv.hpp:49-74) repeats
 * one SpMV on one matrix, so its launches start in whatever the previous one left; this entry says how much of a figure that is.  d_flush: device
 * memory, 16-byte aligned, flush_bytes >= 2 x the Infinity Cache (the bench passes 1 GiB).  No reference counterpart. */
int spmv_acc_time_spmv_cold(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                            const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                            const double *dx, double *dy, const double *d_y0, void *d_flush, long long flush_bytes, float *ms_out);

/* One event pair around all `iters` back-to-back launches (no per-launch markers): *total_ms_out / iters is
 * the average launch duration a solver loop sees. */
int spmv_acc_time_spmv_total(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                             const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                             const double *dx, double *dy, float *total_ms_out);
/* The per-launch protocol once more, with the library's KERNEL CLOCK on (round 5): event_ms_out[i] (may be NULL) is the event pair around call i as
 * above -- the reference harness's figure, which also holds the protocol's floor (two marker packets and the dispatch latency: ~4-7 us on MI355X) --
 * and kernel_ms_out[i] the sum of the durations of the kernels call i launched, each read from the dispatch's own begin / end timestamps
 * (hipExtLaunchKernelGGL start / stop events: what rocprofv3 --kernel-trace reports).  launches_out[i] (may be NULL): kernels per call.
 * replaces: nothing in the reference (its harness has the event pair only, benchmark/utils/timer_utils.h:16-51). */
int spmv_acc_time_spmv_kernels(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                               const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                               const double *dx, double *dy, const double *d_y0, float *event_ms_out, float *kernel_ms_out, int *launches_out);
/* The same region and NOTHING else (round 5): no plan work -- the caller settles the plan first (spmv_acc_prepare_beta) --, no allocation, the
 * event pair is made once per host thread.  A wall clock around this call, between two device synchronisations, reads the K launches' own time
 * (bench.py's `value` / `ms_per_step`: median over repeated regions, the reference's median rule, benchmark/utils/benchmark_time.cpp:23-43). */
int spmv_acc_time_spmv_region(int strategy, int iters, double alpha, double beta, int m, int n, int nnz,
                              const int *h_rowptr, const int *d_rowptr, const int *d_colindex, const double *d_value,
                              const double *dx, double *dy, float *total_ms_out);
/* Streaming-copy ceiling (GB/s, read + write bytes) with the kernels' 16-B non-temporal access shape;
 * replaces: the WITH_MEM_BANDWIDTH macros of src/acc/common/mem_bandwidth.hpp:13-38 as the yardstick. */
double spmv_acc_copy_ceiling_gbs(void *d_dst, const void *d_src, long long bytes, int reps);

/* ---- switches (new) -------------------------------------------------------------------------------------------------
 * A/B knobs for tools/kbench.py ("xcd_chunk", "rowblock_target", "stream_plain", "flat_finish", "flat_npt", ...; the table
 * with every default is in spmv_acc_amd/csrc/config.cpp) and one behavioural switch:
 *   "validate" (0): 1 = check rowptr / colindex of every new matrix on the device before the first launch (rowptr
 *   monotone and non-negative, rowptr[m] == nnz, 0 <= colindex < n); a matrix that fails is refused with
 *   SPMV_ACC_ERR_BAD_ARGUMENT on this and every later call and y is left untouched.  One pass over the indices per
 *   matrix; the reference has no counterpart (its kernels read through whatever the caller passes).
 * Defaults are the shipped configuration; unknown names return -1.  The environment variable
 * SPMV_ACC_TUNABLES="name=value,name=value" seeds the defaults at load time, for executables that cannot call the setter
 * (the reference's spmv-cli / benchmark linked against this library). */
int spmv_acc_set_tunable(const char *name, int value);
int spmv_acc_get_tunable(const char *name);
void spmv_acc_reset_tunables(void);

const char *spmv_acc_version(void);

#ifdef __cplusplus
}
#endif

#endif /* SPMV_ACC_C_ABI_H */
