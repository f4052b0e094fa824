// common/macros.h -- at the reference's include path (src/acc/common/macros.h): unpacks a csr_desc into the local names
// its flat sources use (m, rowptr, colindex, value).
#ifndef SPMV_ACC_AMD_COMMON_MACROS_H
#define SPMV_ACC_AMD_COMMON_MACROS_H

#include "../api/types.h"

#define VAR_FROM_CSR_DESC(d)                                                                                           \
  const int m = (d).rows;                                                                                              \
  const int *rowptr = (d).row_ptr;                                                                                     \
  const int *colindex = (d).col_index;                                                                                 \
  const double *value = (d).values;

#endif
