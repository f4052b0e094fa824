// hip-line/line_strategy.h -- forwarding header at the reference's include path (src/acc/hip-line/line_strategy.h);
// the declarations live in spmv_acc_strategies.hpp.
#ifndef SPMV_ACC_AMD_FWD_HIP_LINE_LINE_STRATEGY_H
#define SPMV_ACC_AMD_FWD_HIP_LINE_LINE_STRATEGY_H
#include "../spmv_acc_strategies.hpp"
#endif
