// flat_locals.hpp -- VAR_FROM_CSR_DESC(desc): declares, in the caller's scope, the locals the reference's flat sources (and its
// benchmark's copy, benchmark/flat/spmv_acc_flat.cpp:82) expect after unpacking a csr_desc<int, double>.
#ifndef SPMV_ACC_AMD_HIP_FLAT_FLAT_LOCALS_HPP
#define SPMV_ACC_AMD_HIP_FLAT_FLAT_LOCALS_HPP

#include "../api/types.h"

#define SPMV_ACC_AMD_BIND(type_, name_, from_) type_ name_ = (from_);
#define VAR_FROM_CSR_DESC(desc_)                                                                                       \
  SPMV_ACC_AMD_BIND(const int, m, (desc_).rows)                /* row count                         */                 \
  SPMV_ACC_AMD_BIND(const int *, rowptr, (desc_).row_ptr)      /* m + 1 offsets                     */                 \
  SPMV_ACC_AMD_BIND(const int *, colindex, (desc_).col_index)  /* column of every non-zero          */                 \
  SPMV_ACC_AMD_BIND(const double *, value, (desc_).values)     /* value of every non-zero           */

#endif
