// hip-adaptive/adaptive.h -- forwarding header at the reference's include path (src/acc/hip-adaptive/adaptive.h);
// the declarations live in spmv_acc_strategies.hpp.
#ifndef SPMV_ACC_AMD_FWD_HIP_ADAPTIVE_ADAPTIVE_H
#define SPMV_ACC_AMD_FWD_HIP_ADAPTIVE_ADAPTIVE_H
#include "../spmv_acc_strategies.hpp"
#endif
