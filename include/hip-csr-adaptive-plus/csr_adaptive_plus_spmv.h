// hip-csr-adaptive-plus/csr_adaptive_plus_spmv.h -- forwarding header at the reference's include path (src/acc/hip-csr-adaptive-plus/csr_adaptive_plus_spmv.h);
// the declarations live in spmv_acc_strategies.hpp.
#ifndef SPMV_ACC_AMD_FWD_HIP_CSR_ADAPTIVE_PLUS_CSR_ADAPTIVE_PLUS_SPMV_H
#define SPMV_ACC_AMD_FWD_HIP_CSR_ADAPTIVE_PLUS_CSR_ADAPTIVE_PLUS_SPMV_H
#include "../spmv_acc_strategies.hpp"
#endif
