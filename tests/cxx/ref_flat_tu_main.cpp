// ref_flat_tu_main.cpp -- main() for tests/test_host_logic.py::test_reference_flat_benchmark_tu_builds_against_include.
// The test compiles the REFERENCE's own benchmark/flat/spmv_acc_flat.cpp and benchmark/utils/benchmark_time.cpp, unmodified and from where they
// lie under /root/reference, with -I include (this repo) in place of the reference's src/acc, and links them with this file and libspmv_acc.so.
// This file only names the four entries that TU defines (benchmark/flat/spmv_acc_flat.h:15-27) so that the linker must resolve them and everything
// they pull from include/hip-flat/* (pre_calc_break_point, pre_calc_break_point_v2, FLAT_KERNEL_WRAPPER, FLAT_KERNEL_ONE_PASS_WRAPPER, flat_config.h).
// Nothing runs without a GPU: the calls sit behind a test that is false.
#include "flat/spmv_acc_flat.h"
int main(int argc, char **) {
  if (argc > 100) {
    csr_desc<int, double> h(0, 0, 0, nullptr, nullptr, nullptr), d(0, 0, 0, nullptr, nullptr, nullptr);
    BenchmarkTime bmt;
    flat_sparse_spmv<FLAT_PRE_CALC_BP_KERNEL_VERSION_V1>(0, 1.0, 1.0, h, d, nullptr, nullptr, &bmt);
    flat_sparse_spmv<FLAT_PRE_CALC_BP_KERNEL_VERSION_V2>(0, 1.0, 1.0, h, d, nullptr, nullptr, &bmt);
    adaptive_flat_sparse_spmv<FLAT_PRE_CALC_BP_KERNEL_VERSION_V1>(0, 0, 0, 1.0, 1.0, d, nullptr, nullptr, &bmt);
    segment_sum_flat_sparse_spmv(0, 1.0, 1.0, h, d, nullptr, nullptr, &bmt);
  }
  return 0;
}
