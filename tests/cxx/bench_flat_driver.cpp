// bench_flat_driver.cpp -- uses include/hip-flat/ the way the reference benchmark's private flat copy does
// (benchmark/flat/spmv_acc_flat.cpp:20-71): unpack the descriptor with VAR_FROM_CSR_DESC, allocate and zero a break-point table,
// launch pre_calc_break_point / pre_calc_break_point_v2 <<<1024, 512>>> itself, then expand FLAT_KERNEL_ONE_PASS_WRAPPER /
// FLAT_KERNEL_WRAPPER in a scope that holds trans, alpha, beta, m, n, nnz, rowptr, colindex, value, x, y.
// Output file: bp_len, the v1 table, the v2 table, then one y per wrapper.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "api/types.h"
#include "common/macros.h"
#include "hip-flat/flat_config.h"
#include "hip-flat/spmv_hip_acc_imp.h"

#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);                      \
      std::abort();                                                                                                    \
    }                                                                                                                  \
  } while (0)

template <int R, int REDUCE_OPTION, int REDUCE_VEC_SIZE, int THREADS_PER_BLOCK, bool V2>
static int *one_pass(int trans, const double alpha, const double beta, const csr_desc<int, double> d_csr_desc, const double *x,
                     double *y, int *len_out) {
  VAR_FROM_CSR_DESC(d_csr_desc)
  const int nnz = d_csr_desc.nnz;
  const int n = d_csr_desc.cols;
  const int HIP_BLOCKS = nnz / (R * THREADS_PER_BLOCK) + ((nnz % (R * THREADS_PER_BLOCK) == 0) ? 0 : 1);
  const int break_points_len = HIP_BLOCKS + 1;
  int *break_points;
  CK(hipMalloc((void **)&break_points, break_points_len * sizeof(int)));
  CK(hipMemset(break_points, 0, break_points_len * sizeof(int)));
  if (V2) {
    (pre_calc_break_point_v2<R * THREADS_PER_BLOCK, 0, int>)<<<1024, 512>>>(rowptr, m, break_points, break_points_len);
  } else {
    (pre_calc_break_point<R * THREADS_PER_BLOCK, 0, int>)<<<1024, 512>>>(rowptr, m, break_points, break_points_len);
  }
  FLAT_KERNEL_ONE_PASS_WRAPPER(R, REDUCE_OPTION, REDUCE_VEC_SIZE, HIP_BLOCKS, THREADS_PER_BLOCK);
  *len_out = break_points_len;
  return break_points;
}

template <int R, int REDUCE_OPTION, int REDUCE_VEC_SIZE, int BLOCKS, int THREADS_PER_BLOCK>
static void multi_pass(int trans, const double alpha, const double beta, int m, int n, int nnz, const int *rowptr,
                       const int *colindex, const double *value, const double *x, double *y) {
  FLAT_KERNEL_WRAPPER(R, REDUCE_OPTION, REDUCE_VEC_SIZE, BLOCKS, THREADS_PER_BLOCK);
}

int main(int argc, char **argv) {
  if (argc < 3) return 2;
  FILE *f = std::fopen(argv[1], "rb");
  if (!f) return 3;
  int hdr[3];
  if (std::fread(hdr, sizeof(int), 3, f) != 3) return 4;
  const int m = hdr[0], n = hdr[1], nnz = hdr[2];
  std::vector<int> rp(m + 1), ci(nnz);
  std::vector<double> v(nnz), x(n), y0(m);
  if (std::fread(rp.data(), sizeof(int), m + 1, f) != size_t(m + 1)) return 4;
  if (std::fread(ci.data(), sizeof(int), nnz, f) != size_t(nnz)) return 4;
  if (std::fread(v.data(), sizeof(double), nnz, f) != size_t(nnz)) return 4;
  if (std::fread(x.data(), sizeof(double), n, f) != size_t(n)) return 4;
  if (std::fread(y0.data(), sizeof(double), m, f) != size_t(m)) return 4;
  std::fclose(f);

  CK(hipSetDevice(0));
  var_csr_desc<int, double> d_csr;
  d_csr.rows = m;
  d_csr.cols = n;
  d_csr.nnz = nnz;
  double *dx, *dy;
  CK(hipMalloc((void **)&d_csr.row_ptr, sizeof(int) * (m + 1)));
  CK(hipMalloc((void **)&d_csr.col_index, sizeof(int) * nnz));
  CK(hipMalloc((void **)&d_csr.values, sizeof(double) * nnz));
  CK(hipMalloc((void **)&dx, sizeof(double) * n));
  CK(hipMalloc((void **)&dy, sizeof(double) * m));
  CK(hipMemcpy(d_csr.row_ptr, rp.data(), sizeof(int) * (m + 1), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_csr.col_index, ci.data(), sizeof(int) * nnz, hipMemcpyHostToDevice));
  CK(hipMemcpy(d_csr.values, v.data(), sizeof(double) * nnz, hipMemcpyHostToDevice));
  CK(hipMemcpy(dx, x.data(), sizeof(double) * n, hipMemcpyHostToDevice));

  FILE *out = std::fopen(argv[2], "wb");
  std::vector<double> y(m);
  auto reset = [&]() { CK(hipMemcpy(dy, y0.data(), sizeof(double) * m, hipMemcpyHostToDevice)); };
  int len1 = 0, len2 = 0;
  reset();
  int *bp1 = one_pass<2, FLAT_REDUCE_OPTION_DIRECT, 1, 512, false>(operation_none, 1.0, 1.0, d_csr.as_const(), dx, dy, &len1);
  CK(hipDeviceSynchronize());
  std::vector<double> y1(m);
  CK(hipMemcpy(y1.data(), dy, sizeof(double) * m, hipMemcpyDeviceToHost));
  reset();
  int *bp2 = one_pass<2, FLAT_REDUCE_OPTION_VEC, 4, 512, true>(operation_none, 1.0, 1.0, d_csr.as_const(), dx, dy, &len2);
  CK(hipDeviceSynchronize());
  std::vector<double> y2(m);
  CK(hipMemcpy(y2.data(), dy, sizeof(double) * m, hipMemcpyDeviceToHost));
  reset();
  multi_pass<2, DEFAULT_FLAT_REDUCE_OPTION, 2, 512, 512>(operation_none, 1.0, 1.0, m, n, nnz, d_csr.row_ptr, d_csr.col_index,
                                                         d_csr.values, dx, dy);
  CK(hipDeviceSynchronize());
  std::vector<double> y3(m);
  CK(hipMemcpy(y3.data(), dy, sizeof(double) * m, hipMemcpyDeviceToHost));

  std::vector<int> t1(len1), t2(len2);
  CK(hipMemcpy(t1.data(), bp1, sizeof(int) * len1, hipMemcpyDeviceToHost));
  CK(hipMemcpy(t2.data(), bp2, sizeof(int) * len2, hipMemcpyDeviceToHost));
  std::fwrite(&len1, sizeof(int), 1, out);
  std::fwrite(t1.data(), sizeof(int), len1, out);
  std::fwrite(t2.data(), sizeof(int), len2, out);
  std::fwrite(y1.data(), sizeof(double), m, out);
  std::fwrite(y2.data(), sizeof(double), m, out);
  std::fwrite(y3.data(), sizeof(double), m, out);
  std::fclose(out);
  return (FLAT_ONE_PASS && FLAT_ONE_PASS_ADAPTIVE) ? 0 : 5;
}
