// api/types.h -- C++ CSR descriptors, source-compatible with the reference's src/acc/api/types.h:8-41
// so that cli/ and benchmark/ sources compile against this tree unchanged: same class names, same
// public members in the same order (rows, cols, nnz, row_ptr, col_index, values), same constructors,
// var_csr_desc::as_const().  Layout matters: the descriptors are passed BY VALUE across the library
// boundary (api/spmv.h:20-21).
#ifndef SPMV_ACC_AMD_API_TYPES_H
#define SPMV_ACC_AMD_API_TYPES_H

enum sparse_operation { operation_none = 0, operation_transpose = 1 };

template <typename I, typename T> class csr_desc;

// mutable view: what the readers / device staging fill in (cli/utils.hpp:94-117)
template <typename I, typename T> class var_csr_desc {
public:
  I rows = 0, cols = 0, nnz = 0;
  I *row_ptr = nullptr;   // rows + 1 entries
  I *col_index = nullptr; // nnz entries
  T *values = nullptr;    // nnz entries

  csr_desc<I, T> as_const();
};

// read-only view handed to the SpMV entry points
template <typename I, typename T> class csr_desc {
public:
  const I rows = 0, cols = 0, nnz = 0;
  const I *row_ptr = nullptr;
  const I *col_index = nullptr;
  const T *values = nullptr;

  csr_desc(const I m, const I n, const I nnz_, const I *rp, const I *ci, const T *v)
      : rows(m), cols(n), nnz(nnz_), row_ptr(rp), col_index(ci), values(v) {}

  csr_desc(const var_csr_desc<I, T> d)
      : rows(d.rows), cols(d.cols), nnz(d.nnz), row_ptr(d.row_ptr), col_index(d.col_index), values(d.values) {}
};

template <typename I, typename T> inline csr_desc<I, T> var_csr_desc<I, T>::as_const() { return csr_desc<I, T>(*this); }

#endif // SPMV_ACC_AMD_API_TYPES_H
