// hip-thread-row/thread_row.h -- forwarding header at the reference's include path (src/acc/hip-thread-row/thread_row.h);
// the declarations live in spmv_acc_strategies.hpp.
#ifndef SPMV_ACC_AMD_FWD_HIP_THREAD_ROW_THREAD_ROW_H
#define SPMV_ACC_AMD_FWD_HIP_THREAD_ROW_THREAD_ROW_H
#include "../spmv_acc_strategies.hpp"
#endif
