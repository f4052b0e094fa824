// api/handle.h -- profiling handle of the csr-adaptive-plus entry, source-compatible with the
// reference's src/acc/api/handle.h:8-13 (the class name keeps the reference's spelling).  Times are
// microseconds, filled by csr_adaptive_plus_sparse_spmv<true, ...>.
#ifndef SPMV_ACC_AMD_API_HANDLE_H
#define SPMV_ACC_AMD_API_HANDLE_H

class SpMVAccHanele {
public:
  double profile_analyze_time = 0; // row-block analysis (host form of the preprocessing pass)
  double profile_kernel_time = 0;  // SpMV kernel(s)
  double profile_destroy_time = 0; // releasing the analysis buffers
};

#endif // SPMV_ACC_AMD_API_HANDLE_H
