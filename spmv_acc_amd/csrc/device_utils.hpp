// device_utils.hpp -- wave64 building blocks shared by every kernel (gfx950 / CDNA4 only).
//
// Role in the reference: src/acc/common/ (dpp_reduce.h:13-54, utils.h:38-59, rocm_global_mem_ops.hpp:25-45)
// provides a DPP wave reduction, a shfl_down reduction and inline-asm wide loads, all behind the
// long-removed __HIP_PLATFORM_HCC__ macro.  This file is the gfx950-native counterpart: DPP
// butterflies expressed with compiler builtins (so hipcc schedules and counts them), 16-byte
// non-temporal streaming loads for the once-read CSR arrays, and an XCD-aware block remap.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace spmv_acc {
namespace dev {

constexpr int kWave = 64;
constexpr int kXcds = 8; // MI355X: 8 XCDs, workgroups dealt round-robin (b and b+8 share an L2)

typedef int int4v __attribute__((ext_vector_type(4)));
typedef double double2v __attribute__((ext_vector_type(2)));

// ---- cross-lane -------------------------------------------------------------------------------
// One DPP move of a 64-bit value = two v_mov_b32_dpp.  All lanes must be active at the call site.
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v);
  int hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xF, 0xF, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double swizzle_xor16_f64(double v) {
  // ds_swizzle bit mode: and_mask 0x1F, or_mask 0, xor_mask 0x10 -> lane ^ 16 inside each 32-lane half
  int lo = __builtin_amdgcn_ds_swizzle(__double2loint(v), 0x401F);
  int hi = __builtin_amdgcn_ds_swizzle(__double2hiint(v), 0x401F);
  return __hiloint2double(hi, lo);
}

// Butterfly all-reduce inside aligned groups of VEC lanes (VEC = 1,2,4,...,64): every lane of a
// group ends with the group's sum.  quad_perm -> row_half_mirror -> row_mirror stay in the DPP
// network (no LDS traffic); the 32- and 64-lane steps cross DPP rows and go through ds_swizzle /
// ds_bpermute.
template <int VEC> __device__ __forceinline__ double group_sum(double v) {
  static_assert(VEC >= 1 && VEC <= 64 && (VEC & (VEC - 1)) == 0, "VEC must be a power of two <= 64");
  if (VEC >= 2) v += dpp_f64<0xB1>(v);  // quad_perm [1,0,3,2]
  if (VEC >= 4) v += dpp_f64<0x4E>(v);  // quad_perm [2,3,0,1]
  if (VEC >= 8) v += dpp_f64<0x141>(v); // row_half_mirror
  if (VEC >= 16) v += dpp_f64<0x140>(v); // row_mirror
  if (VEC >= 32) v += swizzle_xor16_f64(v);
  if (VEC >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// Same butterfly with a run-time (wave-uniform) group width w in {1,2,4,...,64}.
__device__ __forceinline__ double group_sum_dyn(double v, int w) {
  if (w >= 2) v += dpp_f64<0xB1>(v);
  if (w >= 4) v += dpp_f64<0x4E>(v);
  if (w >= 8) v += dpp_f64<0x141>(v);
  if (w >= 16) v += dpp_f64<0x140>(v);
  if (w >= 32) v += swizzle_xor16_f64(v);
  if (w >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// ---- streaming loads ----------------------------------------------------------------------------
// values / colindex are read exactly once per SpMV, 16 B per lane.  Whether they should be non-temporal is a
// per-matrix question on MI355X (measured, kernels.hpp kStreamPolicy*): the templated forms below let the tile
// kernels carry both and the engine picks by timing.  The untemplated forms (non-temporal) serve the scalar paths.
__device__ __forceinline__ int load_stream(const int *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ double load_stream(const double *p) { return __builtin_nontemporal_load(p); }
// NT = false: default cache policy, for matrices small enough to stay in the 256 MB Infinity Cache between SpMVs.
// The vector types are UNDER-aligned on purpose (colindex 4-byte, values 8-byte: the elements' own alignment): gfx950 runs
// global memory in unaligned-access mode, hipcc still emits ONE global_load_dwordx4 with the chosen cache policy, and the same
// kernels serve fresh allocations and sub-array views (a row shard cut out of a larger CSR) at the same speed -- a load whose
// 16 bytes straddle two cache lines costs a second line access, nothing else.
typedef int4v int4v_a4 __attribute__((aligned(4)));
typedef double2v double2v_a8 __attribute__((aligned(8)));
template <bool NT> __device__ __forceinline__ int4v load_stream_i4(const int *p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const int4v_a4 *>(p)) : *reinterpret_cast<const int4v_a4 *>(p);
}
template <bool NT> __device__ __forceinline__ double2v load_stream_d2(const double *p) {
  return NT ? __builtin_nontemporal_load(reinterpret_cast<const double2v_a8 *>(p)) : *reinterpret_cast<const double2v_a8 *>(p);
}

// ---- gathers with a 32-bit byte offset (round 6) ------------------------------------------------------------------------------
// x[col] through the scalar-base form of global_load (`global_load_dwordx2 v, v_off, s[x:x+1]`): the address is the wave-uniform
// pointer plus an UNSIGNED 32-bit byte offset held in one VGPR -- one shift per gather instead of sign-extend + 64-bit shift-add,
// and one address register instead of two (eight gathers in flight per lane: the difference between 65-67 and <= 64 VGPRs, i.e.
// between 7 and 8 waves per SIMD, in the row-block kernel's instances).  Valid while 8 * n < 2^32; the launchers pass the
// kernel-uniform flag `x32` only then (kernels.hpp x32_ok), larger x takes the general staging form with its 64-bit addresses.
__device__ __forceinline__ double gather_u32(const double *x, int col) {
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(x) + (static_cast<unsigned>(col) << 3));
}

// ---- hinted gathers -------------------------------------------------------------------------------------------------------
// A gather whose cache policy is chosen per lane.  Written as `cold ? nontemporal_load(p) : *p` the compiler folds the two loads
// into one plain load of a selected address; buffer loads carry the policy as an immediate operand of the builtin, so the two
// calls stay two instructions (same destination register, complementary lanes, one wait at the first use).  Byte offsets are
// 32 bits: the engine uses hints only where x is below 4 GB; out-of-range offsets read 0 instead of faulting.
// (the cold gathers' cache policy is a build-time constant: an immediate operand.  1 = sc0, 2 = nt, 16 = sc1 and their sums exist for A/B builds --
// make EXTRA=-DSPMV_ACC_COLD_AUX=17 OBJ_DIR=build_exp OUT_DIR=../lib_exp, profiles/probes/far_gather_policy_ab.py)
#ifndef SPMV_ACC_COLD_AUX
#define SPMV_ACC_COLD_AUX 2
#endif
struct XGather {
  __amdgpu_buffer_rsrc_t rsrc;
};
__device__ __forceinline__ XGather make_xgather(const double *x, bool used) {
  XGather g;
  // raw buffer over [x, x + 4 GB): stride 0, DATA_FORMAT 32 (word 3 = 0x00020000, the gfx9 untyped-buffer descriptor)
  g.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(used ? x : nullptr), 0, used ? 0xffffffff : 0, 0x00020000);
  return g;
}
__device__ __forceinline__ double gather_hinted(const XGather &g, int col, unsigned cold) {
  typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
  uint2v r;
  const int off = col << 3;
  if (cold) r = __builtin_amdgcn_raw_buffer_load_b64(g.rsrc, off, 0, SPMV_ACC_COLD_AUX); // aux 2: nt
  else r = __builtin_amdgcn_raw_buffer_load_b64(g.rsrc, off, 0, 0);
  return __hiloint2double(static_cast<int>(r.y), static_cast<int>(r.x));
}

// ---- XCD-aware block remap ------------------------------------------------------------------------
// Hardware deals block b to XCD (b mod 8).  Neighbouring row blocks share x[] lines.  (Round 1's first remap gave each XCD one contiguous
// eighth of the grid: -1 ... +4 % time, never the default, removed in round 6.)
// Chunked order: the 8 XCDs walk the grid together in super-chunks of 8*C blocks, each XCD taking C consecutive
// logical blocks of the super-chunk.  Neighbouring row blocks (shared x[] lines at their common edge) then hit the same
// L2 inside a chunk, while all XCDs stay within the same few MB of the streamed arrays (DRAM pages stay hot, which
// the fully contiguous split above gives up).  Bijective; the ragged tail of the grid keeps the identity order.
__device__ __forceinline__ int xcd_chunked_block(int b, int nblocks, int chunk) {
  const int super = kXcds * chunk;
  const int full = (nblocks / super) * super;
  if (b >= full) return b;
  const int s = b / super;
  const int r = b - s * super;
  return s * super + (r % kXcds) * chunk + r / kXcds;
}

// Zigzag: every other SpMV on a plan walks the grid backwards, so that what the previous SpMV streamed last is still in the 256 MB
// Infinity Cache when this one starts there (cacheable streams; nothing to gain where the plan streams non-temporally).  The
// reversal is applied to the DISPATCH index in groups of 8 and keeps b mod 8, i.e. the XCD a block runs on: a matrix small
// enough to live in the per-XCD L2s keeps finding its lines there (a plain nblocks - 1 - b moved every block to another XCD on
// alternate launches: scircuit-sized 4.5 -> 5.2 us).  Bijective; the ragged tail of the grid keeps its place.
__device__ __forceinline__ int zigzag_block(int b, int nblocks) {
  const int full = nblocks & ~(kXcds - 1);
  return b < full ? full - kXcds - (b & ~(kXcds - 1)) + (b & (kXcds - 1)) : b;
}

// Sum src[k0 .. k1) per lane, for the fix-up kernels (one lane = one cut / sliced row, its carries contiguous in src).
// Ranges of up to 64 entries are added by the lane itself in index order; a longer range -- a row of millions of non-zeros
// cut into thousands of tiles or slices -- would be thousands of dependent loads in one lane (a 2 M-non-zero row: ~80 us),
// so the wave takes such lanes one at a time (ballot order) and sums the range with all 64 lanes + the DPP butterfly.
// Deterministic; every lane of the wave must call it (lanes without a row pass k0 == k1).
__device__ __forceinline__ double wave_range_sum(const double *__restrict__ src, int k0, int k1) {
  const bool giant = k1 - k0 > kWave;
  double s = 0.0;
  if (!giant)
    for (int k = k0; k < k1; ++k) s += src[k];
  unsigned long long todo = __ballot(giant);
  const int lane = threadIdx.x & (kWave - 1);
  while (todo) { // wave-uniform
    const int owner = __ffsll(static_cast<long long>(todo)) - 1;
    todo &= todo - 1;
    const int a = __shfl(k0, owner, kWave), b = __shfl(k1, owner, kWave);
    double part = 0.0;
    for (int k = a + lane; k < b; k += kWave) part += src[k];
    part = group_sum<64>(part);
    if (lane == owner) s = part;
  }
  return s;
}

// Stale-plan guard: called by every SpMV kernel; only the first wave of block 0 does anything (64 4-byte loads, one
// compare).  `guard` holds the rowptr samples of plan-build time, `stale` is a sticky flag in pinned host memory that the
// engine reads before the next call on the plan and in spmv_acc_last_error().
__device__ __forceinline__ void check_plan_guard(const int *__restrict__ rp, int m, const int *__restrict__ guard,
                                                 int *__restrict__ stale) {
  if (guard != nullptr && blockIdx.x == 0 && threadIdx.x < kWave) {
    const int idx = static_cast<int>(static_cast<long long>(threadIdx.x) * m / (kWave - 1));
    if (rp[idx] != guard[threadIdx.x]) __hip_atomic_store(stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// (Round 3 tried a second line of this guard: every workgroup comparing the boundary offsets ITS plan tables were built from -- flat's
// tile digest, the row-block analysis' block ranges, the row digest's bases -- with the live rowptr, which would close the window
// between the 64 samples (for flat exactly).  Two scalar loads per workgroup + one comparison cost 2-3 % on the stream-bound
// stand-ins in all three kernels -- the scalar loads pull the kernels' first lgkmcnt wait forward; as a comparison in flat's row loop,
// where the values are in registers anyway, 0.5-1.5 % -- in-process A/B of two builds, tools/ab_two_libs.py,
// profiles/r03_second_line_guard_ab.txt.  Not kept: the matrices it slows sit within 2 % of the 0.70 gate, and the window it closes
// needs an in-place edit that preserves all 64 samples.)

// y update with the documented semantics y = alpha*A*x + beta*y (api/spmv.h:14).  beta == 0 does
// not read y (BLAS convention; for finite y it equals the reference's alpha*s + 0*y).
// yin: where the OLD y is read.  The reference's entries update y in place (yin == y, what every in-place entry passes); the
// out-of-place entry (spmv_acc_csr_spmv_oop: y_out = alpha*A*x + beta*y_in) passes another vector, which saves an iteration
// that keeps both vectors (and the row-sharded step, whose old slice and new slice live in different buffers) a copy of y per
// SpMV.  Neither pointer is __restrict__ in the kernels: they may be the same vector.
// An EMPTY row under beta == 1 updated in place keeps its value (y + alpha * 0): the row-block-plus kernel neither reads nor
// writes it (keeps_y).  That is 16 B per empty row -- a large share of the traffic of the slabs of the opt-in column-slab blocking on
// power-law matrices (which that kernel runs), whose S passes over y otherwise cost as much as the matrix.  (The one observable
// difference: a y of -0.0 stays -0.0 where alpha * 0 + y would have made it +0.0.)  NOT in the row-block and flat kernels: the test
// sits in their row loop and cost 1.3-3 % on every stream-bound stand-in (in-process A/B of two builds, tools/ab_two_libs.py,
// profiles/r03_keeps_y_ab.txt); in the row-block-plus kernel it measures nothing (+-0.3 %).
__device__ __forceinline__ bool keeps_y(const double *y, const double *yin, double beta) { return beta == 1.0 && yin == y; }

__device__ __forceinline__ void store_y(double *y, const double *yin, int row, double alpha, double beta, double s) {
  // (plain store: non-temporal y stores pay in the row-block kernel only -- flat / row-block-plus / vector-row: -0.6 ... +1.3 %, mixed signs,
  // profiles/r05_short_row_dissection.txt)
  y[row] = (beta == 0.0) ? alpha * s : alpha * s + beta * yin[row];
}

} // namespace dev
} // namespace spmv_acc
