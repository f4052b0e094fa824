// api_driver.cpp -- exercises the C++ surface exactly the way the reference's cli/main.cpp:88-118 and
// benchmark/benchmark_spmv_acc.hpp:27-200 do: include the headers at the reference's paths, build
// csr_desc views, call sparse_csr_spmv and the per-strategy wrappers.  Writes one y per call.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "api/handle.h"
#include "api/spmv.h"
#include "api/types.h"
#include "hip-adaptive/adaptive.h"
#include "hip-csr-adaptive-plus/csr_adaptive_plus_spmv.h"
#include "hip-flat/spmv_hip_acc_imp.h"
#include "hip-line-enhance/line_enhance_spmv.h"
#include "hip-line/line_strategy.h"
#include "hip-vector-row/vector_row.h"
#include "hip/spmv_hip_acc_imp.h"

#define CK(x)                                                                                                          \
  do {                                                                                                                 \
    hipError_t e_ = (x);                                                                                               \
    if (e_ != hipSuccess) {                                                                                            \
      std::fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);                      \
      std::abort();                                                                                                    \
    }                                                                                                                  \
  } while (0)

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
  var_csr_desc<int, double> h_csr, d_csr;
  h_csr.rows = d_csr.rows = m;
  h_csr.cols = d_csr.cols = n;
  h_csr.nnz = d_csr.nnz = nnz;
  h_csr.row_ptr = rp.data();
  h_csr.col_index = ci.data();
  h_csr.values = v.data();
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
  const double alpha = 1.0, beta = 1.0; // cli/main.cpp:95-96
  auto reset = [&]() { CK(hipMemcpy(dy, y0.data(), sizeof(double) * m, hipMemcpyHostToDevice)); };
  auto dump = [&]() {
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(y.data(), dy, sizeof(double) * m, hipMemcpyDeviceToHost));
    std::fwrite(y.data(), sizeof(double), m, out);
  };
  const int trans = operation_none;

  reset(); sparse_csr_spmv(trans, alpha, beta, h_csr.as_const(), d_csr.as_const(), dx, dy); dump();
  reset(); sparse_spmv(trans, alpha, beta, m, n, d_csr.row_ptr, d_csr.col_index, d_csr.values, dx, dy); dump();
  reset(); default_sparse_spmv(trans, alpha, beta, d_csr.as_const(), dx, dy); dump();
  reset(); adaptive_sparse_spmv(trans, alpha, beta, h_csr.as_const(), d_csr.as_const(), dx, dy); dump();
  reset(); flat_sparse_spmv(trans, alpha, beta, h_csr.as_const(), d_csr.as_const(), dx, dy); dump();
  reset(); segment_sum_flat_sparse_spmv(trans, alpha, beta, h_csr.as_const(), d_csr.as_const(), dx, dy); dump();
  reset(); line_enhance_sparse_spmv(trans, alpha, beta, d_csr.as_const(), dx, dy); dump();
  reset(); adaptive_enhance_sparse_spmv(trans, alpha, beta, d_csr.as_const(), dx, dy); dump();
  reset(); adaptive_line_sparse_spmv(trans, alpha, beta, d_csr.as_const(), dx, dy); dump();
  reset(); vec_row_sparse_spmv(trans, alpha, beta, d_csr.as_const(), dx, dy); dump();
  reset(); adaptive_vec_row_sparse_spmv(rp[m / 2], nnz - rp[m / 2], trans, alpha, beta, d_csr.as_const(), dx, dy); dump();
  SpMVAccHanele handle;
  reset(); csr_adaptive_plus_sparse_spmv<true, int, double>(&handle, trans, alpha, beta, h_csr.as_const(), d_csr.as_const(), dx, dy); dump();
  if (handle.profile_kernel_time <= 0.0) return 5;
  reset(); csr_adaptive_plus_sparse_spmv<false, int, double>(nullptr, trans, alpha, beta, h_csr.as_const(), d_csr.as_const(), dx, dy); dump();
  std::fclose(out);
  return 0;
}
