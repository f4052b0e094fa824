// api/spmv.h -- the two public C++ entry points, source-compatible with the reference's
// src/acc/api/spmv.h:20-28.  Both are exported from libspmv_acc.so with C++ linkage (the symbols
// spmv-cli links against); the ten-argument form is ALSO exported with C linkage, see ../spmv_acc.h.
#ifndef SPMV_ACC_AMD_API_SPMV_H
#define SPMV_ACC_AMD_API_SPMV_H

#include "building_config.h"
#include "types.h"

// y = alpha*A*x + beta*y with the active KERNEL_STRATEGY.  h_csr_desc.row_ptr is a HOST array (the
// pickers sample it); every pointer of d_csr_desc, dx and dy is a device pointer.  trans: only
// operation_none.  Asynchronous on the library stream.
void sparse_csr_spmv(int trans, const double alpha, const double beta, const csr_desc<int, double> h_csr_desc,
                     const csr_desc<int, double> d_csr_desc, const double *dx, double *dy);

// Ten-argument form; all pointers are device pointers.  Unlike the reference (api/spmv_imp.cpp:14) it
// never dereferences rowptr on the host.
#ifndef SPMV_ACC_C_ABI_H // (spmv_acc.h, if included first, has declared the C-linkage twin of the same name)
void sparse_spmv(int htrans, const double halpha, const double hbeta, int hm, int hn, const int *rowptr,
                 const int *colindex, const double *value, const double *x, double *y);
#endif

#endif // SPMV_ACC_AMD_API_SPMV_H
