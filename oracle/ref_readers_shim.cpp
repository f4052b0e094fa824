// ref_readers_shim.cpp -- C entry point onto the REFERENCE's own matrix readers.  TEST INFRASTRUCTURE ONLY.
//
// oracle/Makefile compiles this file against the reference's header-only readers where they lie
// (/root/reference/cli/csr_mtx_reader.hpp, csr_binary_reader.hpp, matrix_market_reader.hpp, sparse_format.h, with the
// reference's include roots cli/ and src/acc/ -- none of them needs a generated header) into
// oracle/_ref/libref_readers.so.  Tests use it to pin spmv_acc_amd/cli/matrix_io.hpp: same file in, same CSR out.
#include <cstdlib>
#include <cstring>
#include <string>

#include "csr_binary_reader.hpp"
#include "csr_mtx_reader.hpp"
#include "matrix_market_reader.hpp"

namespace {
template <typename T> T *dup(const T *src, size_t n) {
  T *p = static_cast<T *>(std::malloc(sizeof(T) * (n ? n : 1)));
  if (n) std::memcpy(p, src, sizeof(T) * n);
  return p;
}
} // namespace

// format: 0 = .csr text, 1 = bin2, 2 = MatrixMarket (as cli/main.cpp:47-86 dispatches).  Arrays are malloc'ed copies
// (free with ref_free).  x / xlen are filled for the .csr format only.  Returns 0, or -1 on an exception.
extern "C" int ref_read_matrix(const char *path, int format, int *rows, int *cols, int *nnz, int **rowptr, int **colidx,
                               double **values, double **x, int *xlen) {
  try {
    *x = nullptr;
    *xlen = 0;
    if (format == 0) {
      csr_mtx_reader<int, double> r{std::string(path)};
      r.fill_mtx();
      r.close_stream();
      *rows = r.rows();
      *cols = r.cols();
      *nnz = r.nnz();
      *rowptr = dup(r.csr_indptr.data(), r.csr_indptr.size());
      *colidx = dup(r.csr_indices.data(), r.csr_indices.size());
      *values = dup(r.csr_data.data(), r.csr_data.size());
      *x = dup(r.dense_vector.data(), r.dense_vector.size());
      *xlen = static_cast<int>(r.dense_vector.size());
    } else if (format == 1) {
      csr_binary_reader<int32_t, double> r;
      r.load_mat(std::string(path));
      *rows = r.rows();
      *cols = r.cols();
      *nnz = r.nnz();
      double *v;
      int *ci, *rp;
      r.as_raw_ptr(v, ci, rp);
      *rowptr = dup(rp, static_cast<size_t>(*rows) + 1);
      *colidx = dup(ci, static_cast<size_t>(*nnz));
      *values = dup(v, static_cast<size_t>(*nnz));
    } else {
      matrix_market_reader<int, double> reader;
      matrix_market<int, double> mm = reader.load_mat(std::string(path));
      csr_mtx<int, double> c = mm.to_csr();
      *rows = c.rows;
      *cols = c.cols;
      *nnz = c.nnz;
      *rowptr = dup(c.row_ptr, static_cast<size_t>(c.rows) + 1);
      *colidx = dup(c.col_index, static_cast<size_t>(c.nnz));
      *values = dup(c.values, static_cast<size_t>(c.nnz));
    }
    return 0;
  } catch (...) {
    return -1;
  }
}

extern "C" void ref_free(void *p) { std::free(p); }
