/*
 * spmv_oracle.c -- CPU restatement of the hpcde/spmv-acc hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's
 * library (oracle/liboracle.so).  Nothing under spmv_acc_amd/ links, imports or calls it: the
 * product path is the HIP library and fails loudly when that library is missing.
 *
 * PIN STATUS (also recorded in DESIGN.md):
 *   - oracle_adaptive_plus_analyze : PINNED -- checked bit-for-bit against the reference's own
 *     translation unit src/acc/hip-csr-adaptive-plus/csr_adaptive_plus_analyze.cpp, compiled
 *     unmodified from /root/reference into oracle/_ref/ (see oracle/Makefile, oracle/ref_shim.cpp);
 *     reference-generated fixtures are committed under tests/golden/.
 *   - oracle_host_spmv / verify / verify_y / break points / strategy pickers : PARITY UNPINNED
 *     against reference-run outputs.  The reference ships no golden vectors (no tests, example
 *     matrices are Git-LFS stubs) and the translation units that hold these functions include the
 *     CMake-generated header building_config.h, so they cannot be compiled here without writing a
 *     stand-in for generated code.  They are line-by-line restatements of the cited reference
 *     lines, cross-checked in tests/ against an independent implementation (scipy.sparse) and
 *     against closed-form cases.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off, so products and sums round separately,
 * exactly like the reference's `y0 += value[j] * x[colindex[j]]` compiled without FMA).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------ */
/* y = alpha*A*x + beta*y, sequential, left-to-right accumulation.                              */
/* follows cli/verification.cpp:56-66                                                           */
void oracle_host_spmv(double alpha, double beta, const double *value, const int *rowptr,
                      const int *colindex, int m, int n, int nnz, const double *x, double *y) {
  (void)n;
  (void)nnz;
  for (int i = 0; i < m; i++) {
    double y0 = 0;
    for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
      y0 += value[j] * x[colindex[j]];
    }
    y[i] = alpha * y0 + beta * y[i];
  }
}

/* y = A*x (beta-less overload).  follows cli/verification.cpp:68-78 */
void oracle_host_spmv_plain(const double *value, const int *rowptr, const int *colindex, int m,
                            int n, int nnz, const double *x, double *y) {
  (void)n;
  (void)nnz;
  for (int i = 0; i < m; i++) {
    double y0 = 0;
    for (int j = rowptr[i]; j < rowptr[i + 1]; j++) {
      y0 += value[j] * x[colindex[j]];
    }
    y[i] = y0;
  }
}

/* Same arithmetic per row as oracle_host_spmv, rows distributed over `threads` OpenMP threads on
 * nnz-balanced contiguous row ranges (SURVEY.md 8d "CPU baseline beside it").  Per-row results are
 * bit-identical to the sequential form because each row is still summed left to right.          */
void oracle_host_spmv_omp(double alpha, double beta, const double *value, const int *rowptr,
                          const int *colindex, int m, const double *x, double *y, int threads) {
#ifdef _OPENMP
  if (threads < 1) threads = 1;
#pragma omp parallel num_threads(threads)
  {
    const int t = omp_get_thread_num();
    const int nt = omp_get_num_threads();
    const int64_t nnz = rowptr[m];
    /* row range [r0, r1) holding nnz share [t*nnz/nt, (t+1)*nnz/nt) */
    int64_t lo_target = nnz * t / nt, hi_target = nnz * (t + 1) / nt;
    int r0, r1;
    {
      int lo = 0, hi = m;
      while (lo < hi) { int mid = lo + (hi - lo) / 2; if (rowptr[mid] < lo_target) lo = mid + 1; else hi = mid; }
      r0 = (t == 0) ? 0 : lo;
      lo = 0; hi = m;
      while (lo < hi) { int mid = lo + (hi - lo) / 2; if (rowptr[mid] < hi_target) lo = mid + 1; else hi = mid; }
      r1 = (t == nt - 1) ? m : lo;
    }
    for (int i = r0; i < r1; i++) {
      double y0 = 0;
      for (int j = rowptr[i]; j < rowptr[i + 1]; j++) y0 += value[j] * x[colindex[j]];
      y[i] = alpha * y0 + beta * y[i];
    }
  }
#else
  (void)threads;
  oracle_host_spmv(alpha, beta, value, rowptr, colindex, m, 0, 0, x, y);
#endif
}

/* ---- the CPU baseline's timed form (bench.py `cpu_baseline`, round 5) -------------------------------------------------------
 * Same arithmetic per row as oracle_host_spmv (cli/verification.cpp:56-66), same nnz-balanced row ranges as oracle_host_spmv_omp,
 * but the five arrays are first COPIED into 64-byte-aligned buffers by the very threads that will read them (first touch: on a
 * multi-socket / multi-CCD host a page lands in the memory of the thread that writes it first; numpy-allocated arrays are all
 * touched by one thread, which made this baseline swing by 2x from box to box).  x is touched by the thread whose rows lie on
 * that part of the diagonal (columns [r0*n/m, r1*n/m)).  Then `reps` timed runs, y restored from y0 before each, outside the
 * timed interval; seconds per run into secs[0 .. reps).  The last run's y is copied to y_out (may be NULL) so that the caller can
 * check it against the sequential form.  Threads are pinned by the caller's environment (OMP_PROC_BIND=close, OMP_PLACES=cores:
 * libgomp reads them when it is loaded).  Returns 0, or -1 when a buffer could not be allocated.                                   */
static void nnz_balanced_range(const int *rowptr, int m, int t, int nt, int *r0_out, int *r1_out) {
  const int64_t first = rowptr[0], nnz = (int64_t)rowptr[m] - first;
  const int64_t lo_target = first + nnz * t / nt, hi_target = first + nnz * (t + 1) / nt;
  int lo = 0, hi = m;
  while (lo < hi) { int mid = lo + (hi - lo) / 2; if (rowptr[mid] < lo_target) lo = mid + 1; else hi = mid; }
  *r0_out = (t == 0) ? 0 : lo;
  lo = 0; hi = m;
  while (lo < hi) { int mid = lo + (hi - lo) / 2; if (rowptr[mid] < hi_target) lo = mid + 1; else hi = mid; }
  *r1_out = (t == nt - 1) ? m : lo;
}

static void *alloc64(size_t bytes) { return aligned_alloc(64, (bytes + 63) / 64 * 64 + 64); }

int oracle_host_spmv_bench(double alpha, double beta, const double *value, const int *rowptr, const int *colindex, int m, int n,
                           const double *x, const double *y0, double *y_out, int threads, int reps, double *secs) {
  const int64_t nnz = rowptr[m];
  double *v2 = (double *)alloc64(sizeof(double) * (size_t)nnz), *x2 = (double *)alloc64(sizeof(double) * (size_t)n);
  double *y2 = (double *)alloc64(sizeof(double) * (size_t)m);
  int *c2 = (int *)alloc64(sizeof(int) * (size_t)nnz), *rp2 = (int *)alloc64(sizeof(int) * ((size_t)m + 1));
  int rc = 0;
  if (!v2 || !x2 || !y2 || !c2 || !rp2) rc = -1;
  if (threads < 1) threads = 1;
  if (rc == 0) {
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
#ifdef _OPENMP
      const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
      const int t = 0, nt = 1;
#endif
      int r0, r1;
      nnz_balanced_range(rowptr, m, t, nt, &r0, &r1);
      /* first touch, by the thread that will read them */
      memcpy(rp2 + r0, rowptr + r0, sizeof(int) * (size_t)(r1 - r0 + (t == nt - 1 ? 1 : 0)));
      memcpy(v2 + rowptr[r0], value + rowptr[r0], sizeof(double) * (size_t)(rowptr[r1] - rowptr[r0]));
      memcpy(c2 + rowptr[r0], colindex + rowptr[r0], sizeof(int) * (size_t)(rowptr[r1] - rowptr[r0]));
      memcpy(y2 + r0, y0 + r0, sizeof(double) * (size_t)(r1 - r0));
      const int64_t c_lo = (t == 0) ? 0 : (int64_t)r0 * n / (m > 0 ? m : 1), c_hi = (t == nt - 1) ? n : (int64_t)r1 * n / (m > 0 ? m : 1);
      if (c_hi > c_lo) memcpy(x2 + c_lo, x + c_lo, sizeof(double) * (size_t)(c_hi - c_lo));
      for (int rep = 0; rep < reps; rep++) {
#ifdef _OPENMP
#pragma omp barrier
#endif
        if (rep > 0) memcpy(y2 + r0, y0 + r0, sizeof(double) * (size_t)(r1 - r0)); /* y restored outside the timed interval */
#ifdef _OPENMP
#pragma omp barrier
        const double t0 = omp_get_wtime();
#else
        const double t0 = 0.0;
#endif
        for (int i = r0; i < r1; i++) {
          double acc = 0;
          for (int j = rp2[i]; j < rp2[i + 1]; j++) acc += v2[j] * x2[c2[j]];
          y2[i] = alpha * acc + beta * y2[i];
        }
#ifdef _OPENMP
#pragma omp barrier
        if (t == 0) secs[rep] = omp_get_wtime() - t0;
#else
        secs[rep] = t0;
#endif
      }
      if (y_out) memcpy(y_out + r0, y2 + r0, sizeof(double) * (size_t)(r1 - r0));
    }
  }
  free(v2); free(x2); free(y2); free(c2); free(rp2);
  return rc;
}

/* STREAM triad a[i] = b[i] + s * c[i] over three arrays of `elems` doubles, first-touched and swept by the same static partition
 * (the yardstick the CPU baseline is quoted against: what THIS host's memory delivers to `threads` pinned threads).  Returns the
 * best of `reps` sweeps in GB/s, counting 24 bytes per element (STREAM's convention), or -1.                                      */
double oracle_stream_triad_gbs(long long elems, int threads, int reps) {
  double *a = (double *)alloc64(sizeof(double) * (size_t)elems), *b = (double *)alloc64(sizeof(double) * (size_t)elems);
  double *c = (double *)alloc64(sizeof(double) * (size_t)elems);
  double best = -1.0;
  if (a && b && c && elems > 0) {
    if (threads < 1) threads = 1;
    double best_s = 1e30;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads)
#endif
    {
#ifdef _OPENMP
      const int t = omp_get_thread_num(), nt = omp_get_num_threads();
#else
      const int t = 0, nt = 1;
#endif
      const long long i0 = elems * t / nt, i1 = elems * (t + 1) / nt;
      for (long long i = i0; i < i1; i++) { a[i] = 0.0; b[i] = 1.0; c[i] = 2.0; }
      for (int rep = 0; rep < reps; rep++) {
#ifdef _OPENMP
#pragma omp barrier
        const double t0 = omp_get_wtime();
#else
        const double t0 = 0.0;
#endif
        for (long long i = i0; i < i1; i++) a[i] = b[i] + 3.0 * c[i];
#ifdef _OPENMP
#pragma omp barrier
        if (t == 0) { const double dt = omp_get_wtime() - t0; if (dt < best_s) best_s = dt; }
#else
        (void)t0;
#endif
      }
    }
    if (best_s < 1e29 && best_s > 0.0) best = 24.0 * (double)elems / best_s / 1e9;
    if (a[elems / 2] != 7.0) best = -1.0; /* (keeps the sweep from being optimised away) */
  }
  free(a); free(b); free(c);
  return best;
}

int oracle_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ------------------------------------------------------------------------------------------ */
/* CLI verdict: index of first element with |dy-hy|/|hy| >= 1e-7, or -1 if all pass.            */
/* follows cli/verification.cpp:43-54 (no zero guard: 0/0 = NaN passes, x/0 = inf fails).       */
int oracle_verify(const double *dy, const double *hy, int n) {
  for (int i = 0; i < n; i++) {
    if (fabs(dy[i] - hy[i]) / fabs(hy[i]) >= 1e-7) {
      return i;
    }
  }
  return -1;
}

/* Benchmark verdict.  follows cli/verification.cpp:15-38 */
void oracle_verify_y(const double *dy, const double *hy, int n, double *max_error,
                     int *first_failed_at, int *failed_count) {
  int first = -1;
  int failed = 0;
  double maxe = 0.0;
  for (int i = 0; i < n; i++) {
    const double d = fabs(dy[i] - hy[i]);
    if (d > maxe) maxe = d;
    if ((fabs(hy[i]) <= 1e-12 && d >= 1e-14) || (fabs(hy[i]) > 1e-12 && d / fabs(hy[i]) >= 1e-7)) {
      if (failed <= 0) first = i;
      failed++;
    }
  }
  *max_error = maxe;
  *first_failed_at = first;
  *failed_count = failed;
}

/* The reference's vector generator: libc rand() on a 100-point grid in [-1, 0.96].             */
/* follows cli/utils.hpp:46-56.  (Caller seeds with srand(); the reference never seeds = 1.)    */
void oracle_rand_vector(int n, double *x) {
  for (int i = 0; i < n; i++) {
    x[i] = -1.0 + (1.0 - (-1.0)) * (double)(rand() % 100) / (double)(101);
  }
}
void oracle_srand(unsigned seed) { srand(seed); }

/* Metric definitions of the reference harness.  follows benchmark/utils/statistics_logger.cpp:43-49 */
double oracle_ref_mem_bytes(int rows, int nnz) {
  return (double)(sizeof(double) * (2 * (size_t)rows + (size_t)nnz) + sizeof(int) * ((size_t)rows + 1 + (size_t)nnz));
}
double oracle_ref_gibps(int rows, int nnz, double t_us) {
  return oracle_ref_mem_bytes(rows, nnz) / (1024.0 * 1024.0 * 1024.0) / (t_us / 1e3 / 1e3);
}
double oracle_ref_gflops(int nnz, double t_us) { return (double)(2.0 * nnz) / t_us / 1e3; }

/* ------------------------------------------------------------------------------------------ */
/* Row-block preprocessing pass, DEVICE form (flat strategy): break points.                     */
/* follows src/acc/hip-flat/flat_imp.inl:108-131 with the grid-stride loop run sequentially;    */
/* `break_points` must be pre-zeroed by the caller exactly as flat.cpp:39-40 does (hipMemset).   */
void oracle_break_points(const int *row_ptr, int m, int break_stride, int *break_points, int bp_len) {
  (void)bp_len;
  break_points[0] = 0;
  for (int i = 0; i < m; i++) {
    if (row_ptr[i] / break_stride != row_ptr[i + 1] / break_stride) {
      for (int j = row_ptr[i] / break_stride + 1; j <= row_ptr[i + 1] / break_stride; j++) {
        break_points[j] = i;
      }
      if (row_ptr[i + 1] % break_stride == 0) {
        break_points[row_ptr[i + 1] / break_stride] += 1;
      }
    }
  }
}

/* Length of the break-point array for the one-pass flat kernel.  follows flat.cpp:35-38        */
int oracle_break_points_len(int nnz, int break_stride) {
  const int blocks = nnz / break_stride + ((nnz % break_stride == 0) ? 0 : 1);
  return blocks + 1;
}

/* The simpler v2 formulation (benchmark only).  follows flat_imp.inl:135-152, rows visited in   */
/* ascending order (on the GPU the empty-leading-row case is a write race).       */
void oracle_break_points_v2(const int *row_ptr, int m, int break_stride, int *break_points, int bp_len) {
  (void)bp_len;
  for (int i = 0; i < m; i++) {
    int p1 = row_ptr[i] / break_stride;
    if (row_ptr[i] % break_stride != 0) p1++;
    const int p2 = (row_ptr[i + 1] - 1) / break_stride;
    for (int j = p1; j <= p2; j++) break_points[j] = i;
  }
}

/* ------------------------------------------------------------------------------------------ */
/* Row-block preprocessing pass, HOST form (csr-adaptive-plus analysis).                        */
/* follows src/acc/hip-csr-adaptive-plus/csr_adaptive_plus_analyze.cpp:13-98                    */
/*   break_points      : out, capacity bp_cap, block b owns rows [bp[b], bp[b+1])                */
/*   first_block_of_row: out, m+1 entries, caller pre-zeroed (std::vector::resize does, :27 of   */
/*                       csr_adaptive_plus_spmv.cpp); value = 2*block + long_row_flag             */
/* returns number of blocks (= break_points length - 1), or -1 if bp_cap is too small.           */
int oracle_adaptive_plus_analyze(int m, int nnz, int min_nnz_per_block, int threads_per_block,
                                 int vec_size, const int *host_row_ptr, int *break_points, int bp_cap,
                                 int *first_block_of_row) {
  (void)nnz;
  const int NN_EI = 2; /* csr_adaptive_plus_config.h:12 */
  int bp_size = 0;
  int nnz_count = 0;
  int row_count_i = 0;
#define BP_PUSH(v)                                                                                 \
  do {                                                                                             \
    if (bp_size >= bp_cap) return -1;                                                              \
    break_points[bp_size++] = (v);                                                                 \
  } while (0)

  BP_PUSH(0);
  first_block_of_row[0] = 0;
  const int max_rows_per_block = threads_per_block / vec_size;

  for (int i = 1; i <= m; i++) {
    const int nnz_current_row = host_row_ptr[i] - host_row_ptr[i - 1];
    nnz_count += nnz_current_row;
    row_count_i++;
    if (nnz_count >= min_nnz_per_block) {
      const int multi_blocks = nnz_current_row / min_nnz_per_block;
      const int is_multi_block_row = multi_blocks > 1 ? 1 : 0;
      if (is_multi_block_row) {
        const int new_multi_blocks = nnz_current_row / (NN_EI * min_nnz_per_block);
        for (int k = 0; k < new_multi_blocks; k++) {
          if (k == 0 && nnz_count == nnz_current_row) {
            /* previous block is clean: row i-1 already starts the current block */
          } else {
            BP_PUSH(i - 1);
          }
          if (k == 0) {
            first_block_of_row[i - 1] = (bp_size - 1) * 2 + is_multi_block_row;
          }
        }
        BP_PUSH(i);
      } else {
        BP_PUSH(i);
      }
      nnz_count = 0;
      row_count_i = 0;
    } else if ((row_count_i >= max_rows_per_block) || (i == m)) {
      BP_PUSH(i);
      first_block_of_row[i] = (bp_size - 1) * 2;
      nnz_count = 0;
      row_count_i = 0;
    }
  }
#undef BP_PUSH
  return bp_size - 1;
}

/* Upper bound the reference reserves for the break-point vector.                               */
/* follows csr_adaptive_plus_spmv.cpp:24 (a reserve(), not a hard cap: the vector may outgrow it */
/* when many blocks close on the row limit; callers here pass m+2 which is always enough).       */
int oracle_adaptive_plus_bp_reserve(int nnz, int min_nnz_per_block) {
  return nnz / min_nnz_per_block + ((nnz % min_nnz_per_block) == 0 ? 0 : 1) + 1;
}

/* VEC_SIZE pick of csr_adaptive_plus_sparse_spmv.  follows csr_adaptive_plus_spmv.cpp:139-165  */
int oracle_adaptive_plus_vec(int m, int nnz) {
  const int avg = nnz / m;
  if (avg <= 2) return 1;
  if (avg <= 4) return 2;
  if (avg <= 8) return 4;
  if (avg <= 16) return 8;
  if (avg <= 32) return 16;
  if (avg <= 64) return 32;
  return 64;
}

/* ------------------------------------------------------------------------------------------ */
/* Strategy pickers (host decision logic; integer only).                                        */

/* adaptive strategy decision.  follows src/acc/hip-adaptive/adaptive.cpp:24-66                 */
/* returns 1 vector-row split, 2 adaptive line, 3 adaptive line-enhance, 4 adaptive flat,        */
/* 5 line-enhance (unreachable in the reference, kept for completeness).                         */
int oracle_adaptive_pick(int m, const int *h_row_ptr) {
  const int bp_1 = h_row_ptr[m / 2];
  const int bp_3 = h_row_ptr[m];
  const int avg_nnz_per_row = bp_3 / m;
  const int nnz_block_0 = bp_1 - 0;
  const int nnz_block_1 = bp_3 - bp_1;
  /* the reference divides by the smaller half without a zero guard; a zero
   * half is treated here as "ratio >= 4", which is what an unbounded ratio means. */
  if ((nnz_block_1 > nnz_block_0 && (nnz_block_0 == 0 || nnz_block_1 / nnz_block_0 >= 4)) ||
      (nnz_block_0 > nnz_block_1 && (nnz_block_1 == 0 || nnz_block_0 / nnz_block_1 >= 4))) {
    return 1;
  }
  if (avg_nnz_per_row <= 4) return 2;
  if (bp_3 <= 0xC00000) return 3;
  if (bp_3 > (1 << 23)) return 4;
  return 5;
}

/* adaptive-line parameters.  follows src/acc/hip-line/line_strategy.cpp:52-77 (wavefront 64)   */
void oracle_adaptive_line_params(int m, int nnz, int *vec_size, int *row_num) {
  const int q = nnz / m;
  int v, per_row;
  if (q <= 4) { v = 2; per_row = q + 1; }
  else if (q <= 8) { v = 4; per_row = q + 1; }
  else if (q <= 16) { v = 8; per_row = q + 2; }
  else if (q <= 32) { v = 16; per_row = q + 4; }
  else if (q <= 64) { v = 32; per_row = q + 4; }
  else { v = 64; per_row = q + 4; }
  *vec_size = v;
  *row_num = 512 / per_row; /* BLOCK_LDS_SIZE = HIP_THREADS(256) * R(2), line_strategy.cpp:57-59,37 */
}

/* adaptive line-enhance parameters.  follows src/acc/hip-line-enhance/line_enhance_spmv.cpp:23-69 */
void oracle_adaptive_enhance_params(int m, int nnz, int *rows_per_block, int *vec_size, int *r) {
  const int q = nnz / m;
  if (nnz <= (1 << 24)) {
    if (q >= 32) { *r = 4; *rows_per_block = 64; *vec_size = 8; }
    else { *r = 2; *rows_per_block = 64; *vec_size = 1; }
  } else {
    *r = 2;
    if (q >= 24) { *rows_per_block = 64; *vec_size = 4; }
    else { *rows_per_block = 128; *vec_size = 1; }
  }
}

/* adaptive flat reduce width (1 = direct).  follows src/acc/hip-flat/flat.cpp:47-57,112-130     */
int oracle_adaptive_flat_vec(int m, const int *h_row_ptr) {
  const int bp_1 = h_row_ptr[m / 2];
  const int bp_2 = h_row_ptr[m];
  const int nnz_block_0 = bp_1 - 0;
  const int nnz_block_1 = bp_2 - bp_1;
  const int a0 = 2 * nnz_block_0 / m, a1 = 2 * nnz_block_1 / m;
  const int a = a0 > a1 ? a0 : a1;
  if (a <= 32) return 1;
  if (a <= 64) return 4;
  return 16;
}

/* wavefront split point of the adaptive vector-row kernel.  follows vector_row.cpp:34-36       */
int oracle_adaptive_vec_row_bp(int weight_block_0, int weight_block_1) {
  const int bp = (int)round((16.0 * weight_block_0) / ((double)weight_block_0 + (double)weight_block_1));
  const int lo = bp > 1 ? bp : 1;
  return lo < 15 ? lo : 15;
}
