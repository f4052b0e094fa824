// hip-flat/spmv_hip_acc_imp.h -- forwarding header at the reference's include path (src/acc/hip-flat/spmv_hip_acc_imp.h);
// the declarations live in spmv_acc_strategies.hpp.
#ifndef SPMV_ACC_AMD_FWD_HIP_FLAT_SPMV_HIP_ACC_IMP_H
#define SPMV_ACC_AMD_FWD_HIP_FLAT_SPMV_HIP_ACC_IMP_H
#include "../spmv_acc_strategies.hpp"
#include "flat_compat.hpp" // break-point kernels + launch macros the reference benchmark's flat copy uses (hipcc only)
#endif
