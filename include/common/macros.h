// Forwarding header at the reference's include path src/acc/common/macros.h.  VAR_FROM_CSR_DESC -- the four locals (m, rowptr,
// colindex, value) the reference's flat sources pull out of a csr_desc -- is defined with the other flat compatibility pieces.
#ifndef SPMV_ACC_AMD_FWD_COMMON_MACROS_H
#define SPMV_ACC_AMD_FWD_COMMON_MACROS_H
#include "../hip-flat/flat_locals.hpp"
#endif
