// hip-line-enhance/line_enhance_spmv.h -- forwarding header at the reference's include path (src/acc/hip-line-enhance/line_enhance_spmv.h);
// the declarations live in spmv_acc_strategies.hpp.
#ifndef SPMV_ACC_AMD_FWD_HIP_LINE_ENHANCE_LINE_ENHANCE_SPMV_H
#define SPMV_ACC_AMD_FWD_HIP_LINE_ENHANCE_LINE_ENHANCE_SPMV_H
#include "../spmv_acc_strategies.hpp"
#endif
