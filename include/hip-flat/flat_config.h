// hip-flat/flat_config.h -- the compile-time switches the reference's flat sources and its benchmark's private flat copy
// (benchmark/flat/spmv_acc_flat.cpp:80-126) select their kernel variants with.  This library picks tile shape, lanes per row
// and cut-row handling itself (k_flat.hip), so the values only have to exist with the reference's names and meanings for those
// sources to compile; they are usable as template arguments.
#ifndef SPMV_ACC_AMD_HIP_FLAT_FLAT_CONFIG_H
#define SPMV_ACC_AMD_HIP_FLAT_FLAT_CONFIG_H

enum : int {
  FLAT_REDUCE_OPTION_VEC = 0,                // a vector of lanes per row
  FLAT_REDUCE_OPTION_VEC_MEM_COALESCING = 1, // ... with the results moved through LDS for a coalesced store
  FLAT_REDUCE_OPTION_DIRECT = 2,             // one lane per row
  FLAT_REDUCE_OPTION_SEGMENT_SUM = 3,        // LDS segmented scan
  DEFAULT_FLAT_REDUCE_OPTION = FLAT_REDUCE_OPTION_VEC_MEM_COALESCING
};

constexpr bool FLAT_ONE_PASS = true;          // one workgroup per tile of non-zeros (the only form this library runs)
constexpr bool FLAT_ONE_PASS_ADAPTIVE = true; // reduction variant chosen from the density of the two row halves

#endif
