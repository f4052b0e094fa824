// hip-wf-row/spmv_hip.h -- forwarding header at the reference's include path (src/acc/hip-wf-row/spmv_hip.h);
// the declarations live in spmv_acc_strategies.hpp.
#ifndef SPMV_ACC_AMD_FWD_HIP_WF_ROW_SPMV_HIP_H
#define SPMV_ACC_AMD_FWD_HIP_WF_ROW_SPMV_HIP_H
#include "../spmv_acc_strategies.hpp"
#endif
