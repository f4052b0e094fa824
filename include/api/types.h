// api/types.h -- CSR descriptors for the C++ entry points.
//
// Source- and layout-compatible with the reference's src/acc/api/types.h:8-41, so that the reference's cli/ and
// benchmark/ sources compile against this tree unchanged: class templates `var_csr_desc<I, T>` (mutable, filled by
// readers / device staging) and `csr_desc<I, T>` (read-only, what the SpMV entry points take BY VALUE, api/spmv.h:20-21),
// the six public members in the order rows, cols, nnz, row_ptr, col_index, values, `var_csr_desc::as_const()`, and the
// enum `sparse_operation`.  Both classes here share one field block so the order cannot drift apart.
#ifndef SPMV_ACC_AMD_API_TYPES_H
#define SPMV_ACC_AMD_API_TYPES_H

enum sparse_operation { operation_none = 0, operation_transpose = 1 };

namespace spmv_acc_types {
// The six CSR fields.  Count: I or const I; IndexPtr / ValuePtr: pointers to (const) index / value arrays.
template <typename Count, typename IndexPtr, typename ValuePtr> struct csr_fields {
  Count rows;          // number of matrix rows
  Count cols;          // number of matrix columns
  Count nnz;           // number of stored entries
  IndexPtr row_ptr;    // rows + 1 offsets into col_index / values
  IndexPtr col_index;  // nnz column ids
  ValuePtr values;     // nnz values
};
} // namespace spmv_acc_types

template <typename I, typename T> class csr_desc;

template <typename I, typename T> class var_csr_desc : public spmv_acc_types::csr_fields<I, I *, T *> {
  using fields = spmv_acc_types::csr_fields<I, I *, T *>;

public:
  var_csr_desc() : fields{I(0), I(0), I(0), nullptr, nullptr, nullptr} {}
  csr_desc<I, T> as_const();
};

template <typename I, typename T> class csr_desc : public spmv_acc_types::csr_fields<const I, const I *, const T *> {
  using fields = spmv_acc_types::csr_fields<const I, const I *, const T *>;

public:
  csr_desc(const I m, const I n, const I entries, const I *offsets, const I *columns, const T *vals)
      : fields{m, n, entries, offsets, columns, vals} {}

  csr_desc(const var_csr_desc<I, T> d) : fields{d.rows, d.cols, d.nnz, d.row_ptr, d.col_index, d.values} {}
};

template <typename I, typename T> inline csr_desc<I, T> var_csr_desc<I, T>::as_const() { return csr_desc<I, T>(*this); }

static_assert(sizeof(csr_desc<int, double>) == 3 * sizeof(int) + 4 /*padding*/ + 3 * sizeof(void *),
              "csr_desc<int,double> must keep the reference's layout: it crosses the library boundary by value");

#endif // SPMV_ACC_AMD_API_TYPES_H
