// building_config.h -- hand-maintained counterpart of the header the reference generates with CMake
// from src/building_config.h.in:11-42.  The reference bakes exactly one KERNEL_STRATEGY_* macro into
// the library; here the library carries every strategy and the macro only selects the DEFAULT one
// (override at run time: SPMV_ACC_KERNEL_STRATEGY / spmv_acc_set_strategy, include/spmv_acc.h).
// Build with e.g. -DKERNEL_STRATEGY_FLAT to change the default; nothing defined = ADAPTIVE.
#ifndef SPMV_ACC_AMD_BUILDING_CONFIG_H
#define SPMV_ACC_AMD_BUILDING_CONFIG_H

#define ACCELERATE_ENABLED
#define ARCH_HIP
#define ARCH_NAME "gfx950"

// MI355X: 256 CUs, 64-lane wavefronts (the reference defaults to AVAILABLE_CU 60, config.cmake:12-14)
#ifndef AVAILABLE_CU
#define AVAILABLE_CU 256
#endif
#ifndef __WF_SIZE__
#define __WF_SIZE__ 64
#endif
constexpr int __WRAP_SIZE__ = __WF_SIZE__;

#if !defined(KERNEL_STRATEGY_DEFAULT) && !defined(KERNEL_STRATEGY_ADAPTIVE) && !defined(KERNEL_STRATEGY_THREAD_ROW) && \
    !defined(KERNEL_STRATEGY_WAVEFRONT_ROW) && !defined(KERNEL_STRATEGY_BLOCK_ROW_ORDINARY) &&                         \
    !defined(KERNEL_STRATEGY_LIGHT) && !defined(KERNEL_STRATEGY_VECTOR_ROW) && !defined(KERNEL_STRATEGY_LINE) &&       \
    !defined(KERNEL_STRATEGY_FLAT) && !defined(KERNEL_STRATEGY_LINE_ENHANCE)
#define KERNEL_STRATEGY_ADAPTIVE
#endif

// building_config.h.in:18-21: flavour of the wavefront-row reduction.  One DPP reduction serves all three names here; the
// macros exist so that code written against the generated header compiles.  FLAT_SEGMENT_SUM_REDUCE (:40) and
// DEVICE_SIDE_VERIFY (:37) arrive from the build (-D..., CMakeLists.txt options of the same names as config.cmake:9,51); the
// former makes the segmented-scan reduction flat's default (tunable flat_reduce, config.cpp), as strategy_picker.cpp:34-39 does.
#if !defined(WF_REDUCE_DEFAULT) && !defined(WF_REDUCE_LDS) && !defined(WF_REDUCE_REG)
#define WF_REDUCE_DEFAULT
#endif

#endif // SPMV_ACC_AMD_BUILDING_CONFIG_H
