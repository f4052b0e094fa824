// tile_stage.hpp -- cooperative "multiply into LDS" step shared by the row-block, nnz-tile and
// row-block-plus kernels.
//
// Role in the reference: the `shared_val[i] = csr_val[idx] * x[csr_col_ind[idx]]` loops of
// hip-flat/flat_imp_one_pass.hpp:35-39, hip-line-enhance/line_enhance_spmv_imp.inl:55-62 and
// hip-csr-adaptive-plus/csr_adaptive_plus_spmv_imp.inl:152-160 (one 4- or 8-byte load per lane).
// Here a lane owns 4 consecutive non-zeros per step: one 16-B colindex load, two 16-B value loads
// (cache policy NTC / NTV chosen per matrix by the engine), four x[] gathers, one 32-B LDS store.  Loads of all steps are issued before
// the first gather so every lane keeps NPT/4 * 48 B of stream plus NPT gathers in flight.
#pragma once

#include "device_utils.hpp"
#include "kernels.hpp"

namespace spmv_acc {
namespace dev {

// Stage products of non-zeros [a0, a0 + THREADS*NPT) that lie below `hi` into lds[0 .. THREADS*NPT).
//   a0  : first non-zero of the tile, multiple of 4 (16-B loads of a 16-B-aligned array are then aligned too)
//   hi  : exclusive bound of the non-zeros this block needs (hi <= nnz); groups at or above it are skipped
//   nnz : total non-zeros (array length) -- only the last, ragged group of the arrays takes the scalar path
// Slots of lds whose non-zero index is < a-block's-first-nnz or >= hi hold unspecified values; no
// reader touches them.
//
// HINT (gather hints, tuner.cpp ensure_hint): `cold` holds one bit per non-zero, set where the plan's column census found the
// x line of that non-zero outside the set of hot lines that fit an L2.  Cold gathers are issued non-temporal, so the lines they
// bring do not displace the hot ones (skewed_gather_bench.hip: 66 -> 74-76 G gathers/s on R-MAT columns; non-temporal for ALL
// gathers: 43).  The bits only steer the cache policy: stale or arbitrary bits cannot change a sum.
// The branch-free step of a wavefront whose steps ALL start below `hi` (stage_products' wave-uniform test): no branch at all between the
// loads.  Across the per-step branches of the earlier form hipcc's waitcnt pass took the most conservative count: the first gathers waited at
// vmcnt(2) -- the first step's values AND the second step's colindex back -- where vmcnt(5) is what they need (round 6, found on the 16-bit
// column path, whose first decode waited at vmcnt(0)).
template <int THREADS, int NPT, bool NTC, bool NTV, bool HINT>
__device__ __forceinline__ void stage_products_all(double *__restrict__ lds, int a0, int hi, const int *__restrict__ ci,
                                                   const double *__restrict__ v, const double *__restrict__ x,
                                                   const unsigned char *__restrict__ cold) {
  constexpr int K = NPT / 4;
  int4v c[K];
  double2v va[K], vb[K];
  unsigned nib[K]; // HINT: the cold bits of this lane's four non-zeros
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int j = a0 + 4 * (threadIdx.x + k * THREADS);
    const int jc = (j < hi) ? j : a0; // lanes past hi in the boundary wave re-read the tile's first group (L1 hit)
    c[k] = load_stream_i4<NTC>(ci + jc);
    va[k] = load_stream_d2<NTV>(v + jc);
    vb[k] = load_stream_d2<NTV>(v + jc + 2);
    if (HINT) nib[k] = static_cast<unsigned>(cold[jc >> 3]) >> (jc & 4); // (jc is a multiple of 4; a wave reads 32 consecutive bytes)
  }
  double xg[K][4];
  const XGather xr = make_xgather(x, HINT);
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (HINT) {
      xg[k][0] = gather_hinted(xr, c[k].x, nib[k] & 1u);
      xg[k][1] = gather_hinted(xr, c[k].y, nib[k] & 2u);
      xg[k][2] = gather_hinted(xr, c[k].z, nib[k] & 4u);
      xg[k][3] = gather_hinted(xr, c[k].w, nib[k] & 8u);
    } else { // one VGPR and one shift per gather address (the caller vouches for 8 * n < 2^32: x32)
      xg[k][0] = gather_u32(x, c[k].x);
      xg[k][1] = gather_u32(x, c[k].y);
      xg[k][2] = gather_u32(x, c[k].z);
      xg[k][3] = gather_u32(x, c[k].w);
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int g = threadIdx.x + k * THREADS;
    double2v p0, p1;
    p0.x = va[k].x * xg[k][0];
    p0.y = va[k].y * xg[k][1];
    p1.x = vb[k].x * xg[k][2];
    p1.y = vb[k].y * xg[k][3];
    double2v *dst = reinterpret_cast<double2v *>(lds + 4 * g);
    dst[0] = p0;
    dst[1] = p1;
  }
}

// The wavefronts at a tile's end (their last step starts at or above `hi`; one in eight on a row-block grid): one step at a time, the loop NOT
// unrolled -- with NPT = 8 such a wavefront has at most one step to do anyway, and a body that cannot set the kernel's register count (the
// unrolled form with per-step branches cost up to 16 registers of undefined-value copies in some instances).
template <int THREADS, int NPT, bool NTC, bool NTV, bool HINT>
__device__ __forceinline__ void stage_products_tail(double *__restrict__ lds, int a0, int hi, const int *__restrict__ ci,
                                                    const double *__restrict__ v, const double *__restrict__ x,
                                                    const unsigned char *__restrict__ cold) {
  const XGather xr = make_xgather(x, HINT);
#pragma nounroll
  for (int k = 0; k < NPT / 4 - 1; ++k) {
    const int wave_j = __builtin_amdgcn_readfirstlane(a0 + 4 * ((threadIdx.x & ~(kWave - 1)) + k * THREADS));
    if (wave_j >= hi) break; // wave-uniform
    const int g = threadIdx.x + k * THREADS;
    const int j = a0 + 4 * g;
    const int jc = (j < hi) ? j : a0;
    const int4v c = load_stream_i4<NTC>(ci + jc);
    const double2v va = load_stream_d2<NTV>(v + jc), vb = load_stream_d2<NTV>(v + jc + 2);
    double xg[4];
    if (HINT) {
      const unsigned nib = static_cast<unsigned>(cold[jc >> 3]) >> (jc & 4);
      xg[0] = gather_hinted(xr, c.x, nib & 1u);
      xg[1] = gather_hinted(xr, c.y, nib & 2u);
      xg[2] = gather_hinted(xr, c.z, nib & 4u);
      xg[3] = gather_hinted(xr, c.w, nib & 8u);
    } else {
      xg[0] = gather_u32(x, c.x);
      xg[1] = gather_u32(x, c.y);
      xg[2] = gather_u32(x, c.z);
      xg[3] = gather_u32(x, c.w);
    }
    double2v p0, p1;
    p0.x = va.x * xg[0];
    p0.y = va.y * xg[1];
    p1.x = vb.x * xg[2];
    p1.y = vb.y * xg[3];
    double2v *dst = reinterpret_cast<double2v *>(lds + 4 * g);
    dst[0] = p0;
    dst[1] = p1;
  }
}

template <int THREADS, int NPT, bool NTC = true, bool NTV = true, bool HINT = false>
__device__ __forceinline__ void stage_products(double *__restrict__ lds, int a0, int hi, int nnz,
                                               const int *__restrict__ ci, const double *__restrict__ v,
                                               const double *__restrict__ x, bool allow_fast = true,
                                               const unsigned char *__restrict__ cold = nullptr, bool x32 = false) {
  static_assert(NPT % 4 == 0, "NPT must be a multiple of 4");
  constexpr int K = NPT / 4;
  // Branch-free form (wave-uniform test): a0 is a multiple of 4, so the last 4-group that starts below `hi` ends at
  // round_up(hi, 4); when that is still inside the arrays every group a lane may load is in bounds (the elements
  // between hi and round_up(hi, 4) belong to the next rows: valid entries, valid columns).  Work is skipped per WAVE
  // (scalar branches on a wave-uniform test); inside the one wave that straddles `hi`, lanes past it re-load the
  // tile's first group instead of being masked off (an L1 hit, no extra memory request).  No per-lane branches remain,
  // so hipcc places exact s_waitcnt counts: all stream loads of the wave, then all its gathers in flight.  (With
  // per-lane branches it waited for half of the first step's gathers before issuing the second step's.)  Products of
  // re-loaded groups land in slots >= hi - a0 that no reader touches.
  if (allow_fast && x32 && ((hi + 3) & ~3) <= nnz) {
    // the wavefront's LAST step starts below hi: so do all its steps (wave-uniform)
    const int last_j = __builtin_amdgcn_readfirstlane(a0 + 4 * ((threadIdx.x & ~(kWave - 1)) + (K - 1) * THREADS));
    if (last_j < hi) stage_products_all<THREADS, NPT, NTC, NTV, HINT>(lds, a0, hi, ci, v, x, cold);
    else stage_products_tail<THREADS, NPT, NTC, NTV, HINT>(lds, a0, hi, ci, v, x, cold);
    return;
  }
  // General form: the tile that holds the ragged end of the arrays (one in the grid), `stage_fast = 0` (tests), and x of 4 GB and more
  // (64-bit gather addresses).  One step at a time (the loop is NOT unrolled), so that its two address registers per gather never set the
  // kernel's register count: the branch-free form above decides that (55-59 VGPRs in the row-block kernel's instances, 8 waves per SIMD).
#pragma nounroll
  for (int k = 0; k < K; ++k) {
    const int g = threadIdx.x + k * THREADS;
    const int j = a0 + 4 * g;
    if (j < hi && j + 4 <= nnz) {
      const int4v c = load_stream_i4<NTC>(ci + j);
      const double2v va = load_stream_d2<NTV>(v + j), vb = load_stream_d2<NTV>(v + j + 2);
      double2v p0, p1;
      p0.x = va.x * x[c.x];
      p0.y = va.y * x[c.y];
      p1.x = vb.x * x[c.z];
      p1.y = vb.y * x[c.w];
      double2v *dst = reinterpret_cast<double2v *>(lds + 4 * g);
      dst[0] = p0;
      dst[1] = p1;
    } else if (j < hi) { // ragged end of the arrays (at most one group in the whole grid)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (j + e < nnz) lds[4 * g + e] = v[j + e] * x[ci[j + e]];
      }
    }
  }
}

// ---- split staging -----------------------------------------------------------------------------------------------------------
// The same step in two halves, for kernels that want the tile's stream loads in flight BEFORE their own dependent scalar /
// rowptr loads (flat on small grids: a tile kernel is a chain of round trips -- break point -> rowptr -> stream -> gather ->
// LDS -> y -- and on a grid of one or two workgroups per CU nothing else hides them).  stage_issue starts the 16-B stream
// loads of the branch-free form into registers; stage_finish gathers x, multiplies and writes the tile.  The price is
// NPT/4 * 12 VGPRs held across whatever the caller does in between, which costs occupancy that only small grids can spare.
// Only the branch-free form is split: callers test stage_fast_ok() and use stage_products() otherwise.
__device__ __forceinline__ bool stage_fast_ok(int hi, int nnz) { return ((hi + 3) & ~3) <= nnz; }

template <int NPT> struct StreamRegs {
  int4v c[NPT / 4];
  double2v va[NPT / 4], vb[NPT / 4];
  bool has[NPT / 4]; // wave-uniform
};

template <int THREADS, int NPT, bool NTC, bool NTV>
__device__ __forceinline__ void stage_issue(StreamRegs<NPT> &R, int a0, int hi, const int *__restrict__ ci,
                                            const double *__restrict__ v) {
#pragma unroll
  for (int k = 0; k < NPT / 4; ++k) {
    const int wave_j = __builtin_amdgcn_readfirstlane(a0 + 4 * ((threadIdx.x & ~(kWave - 1)) + k * THREADS));
    R.has[k] = wave_j < hi;
    if (R.has[k]) {
      const int j = a0 + 4 * (threadIdx.x + k * THREADS);
      const int jc = (j < hi) ? j : a0;
      R.c[k] = load_stream_i4<NTC>(ci + jc);
      R.va[k] = load_stream_d2<NTV>(v + jc);
      R.vb[k] = load_stream_d2<NTV>(v + jc + 2);
    }
  }
}

template <int THREADS, int NPT>
__device__ __forceinline__ void stage_finish(double *__restrict__ lds, const StreamRegs<NPT> &R, const double *__restrict__ x) {
  constexpr int K = NPT / 4;
  double xg[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (R.has[k]) {
      xg[k][0] = x[R.c[k].x];
      xg[k][1] = x[R.c[k].y];
      xg[k][2] = x[R.c[k].z];
      xg[k][3] = x[R.c[k].w];
    }
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    if (R.has[k]) {
      const int g = threadIdx.x + k * THREADS;
      double2v p0, p1;
      p0.x = R.va[k].x * xg[k][0];
      p0.y = R.va[k].y * xg[k][1];
      p1.x = R.vb[k].x * xg[k][2];
      p1.y = R.vb[k].y * xg[k][3];
      double2v *dst = reinterpret_cast<double2v *>(lds + 4 * g);
      dst[0] = p0;
      dst[1] = p1;
    }
  }
}

// ---- staging from the 16-bit column encoding (k_col16.hip, kernels.hpp Col16) --------------------------------------------------------
// The branch-free step of stage_products with the colindex stream (4 B per non-zero) replaced by the plan's encoding: per lane and step one
// 8-B load of four 16-bit offsets, per wavefront and step ONE load of the chunk's record (R ints: base, escape count, overflow start, the
// chunk's first R - 4 escaped columns).  The tile origin a0 is a multiple of 256, so a wavefront's step is exactly one chunk and the
// record's address depends on nothing but the step: records, offsets and values are all requested before anything is waited for (records
// first: they come back first and the gathers need only them and the offsets).  Decoding: base from lane 0 of the record (a scalar), column
// = base + offset; in a chunk WITH escapes (wave-uniform test on the record's count) an escaped entry takes its column out of the record by
// rank -- ballots + popcounts for the rank, one ds_bpermute per element slot that holds an escape anywhere in the wavefront; only a chunk
// with more than R - 4 escapes reads the overflow list (a dependent load, <= 1 % of the chunks by the choice of R).
// (A form whose codes NAME the escape -- 0xFF00 + index, no ranks: 16 instead of ~40 vector instructions per four non-zeros, 4 more registers --
// measured the same within 0.5 % on eight stand-ins under pinned plans and 2-4 % slower on the long-row one: profiles/r06_col16_counters.md
// section 5.  The instruction count was not what held this path back; the wait in front of the first decode was: see stage_products_c16_body.)
// (Round 2's form -- base[] and esc_start[] arrays, one escape list -- needed two scalars back before it could ask for the escapes, and those
// before the first gather: profiles/r06_col16_counters.md.)
//   a0  : tile origin, multiple of 256;   lo4 : first group the workgroup needs (multiple of 4, a0 <= lo4);   hi : exclusive bound
// Lanes whose group lies outside [lo4, hi) still load their own offsets (the ranks count every escape of the chunk) but re-read the values of
// group lo4 and gather the chunk's base column (L1 hits, no extra lines); their products land in slots no reader touches.
// Preconditions (checked by the caller): every 4-group below `hi` lies inside the arrays (stage_fast_ok) and 8 * n < 2^32 (x32_ok).
typedef unsigned int uint2v __attribute__((ext_vector_type(2)));
typedef uint2v uint2v_a2 __attribute__((aligned(2)));

// One step's decode: the four columns of this lane's group out of the record (held one entry per lane in rv) and the four codes dq.
__device__ __forceinline__ void c16_decode(int (&c)[4], const uint2v d, int rv, int R, int lane, const int *__restrict__ ovf) {
  const int bs = __builtin_amdgcn_readlane(rv, 0);
  const int nesc = __builtin_amdgcn_readlane(rv, 1);
  const int dq[4] = {static_cast<int>(d.x & 0xFFFFu), static_cast<int>(d.x >> 16), static_cast<int>(d.y & 0xFFFFu), static_cast<int>(d.y >> 16)};
#pragma unroll
  for (int q = 0; q < 4; ++q) c[q] = bs + dq[q];
  if (nesc > 0) { // wave-uniform
    const int E = R - 4;
    const unsigned long long lt = (1ull << lane) - 1ull;
    const bool e[4] = {dq[0] == 0xFFFF, dq[1] == 0xFFFF, dq[2] == 0xFFFF, dq[3] == 0xFFFF};
    const unsigned long long slot[4] = {__ballot(e[0]), __ballot(e[1]), __ballot(e[2]), __ballot(e[3])};
    // rank of this lane's first escape in the chunk: the escapes held by lower lanes (records list them in non-zero order)
    int r = __popcll(slot[0] & lt) + __popcll(slot[1] & lt) + __popcll(slot[2] & lt) + __popcll(slot[3] & lt);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (slot[q]) { // wave-uniform: some lane's q-th entry is an escape
        const int got = __shfl(rv, (4 + r) & (kWave - 1), kWave);
        if (e[q]) c[q] = r < E ? got : ovf[__builtin_amdgcn_readlane(rv, 2) + r - E];
      }
      r += e[q] ? 1 : 0;
    }
  }
}

// A wavefront whose steps ALL start below `hi` (the caller's wave-uniform test): no branch between the loads.  With per-step branches the
// waitcnt pass cannot count across them: it placed s_waitcnt vmcnt(0) in front of the first decode (ALL the value loads back before the first
// gather leaves) where vmcnt(5) is what the decode needs (record + offsets) -- 2-5 % of these kernels' time on the FEM-class stand-ins
// (profiles/r06_col16_counters.md section 5).  Seven of eight wavefronts of a row-block grid take this body (tiles are filled to 1800 of 2048).
template <int THREADS, int NPT, bool NTC, bool NTV>
__device__ __forceinline__ void stage_products_c16_all(double *__restrict__ lds, int a0, int lo4, int hi, const Col16Dev &C,
                                                       const double *__restrict__ v, const double *__restrict__ x) {
  constexpr int K = NPT / 4;
  uint2v d[K];
  double2v va[K], vb[K];
  int rv[K];
  const int lane = threadIdx.x & (kWave - 1);
  const int R = C.rec_ints;
  const int rl = lane < R ? lane : 0; // (lanes past the record re-read its first entry: no branch, no extra line)
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int wave_j = __builtin_amdgcn_readfirstlane(a0 + 4 * ((threadIdx.x & ~(kWave - 1)) + k * THREADS));
    rv[k] = C.rec[static_cast<long long>(wave_j >> 8) * R + rl];
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int j = a0 + 4 * (threadIdx.x + k * THREADS);
    const uint2v_a2 *p = reinterpret_cast<const uint2v_a2 *>(C.d16 + j); // (the offsets are padded to whole chunks)
    d[k] = NTC ? __builtin_nontemporal_load(p) : *p;
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int j = a0 + 4 * (threadIdx.x + k * THREADS);
    const int jv = (j >= lo4 && j < hi) ? j : lo4;
    va[k] = load_stream_d2<NTV>(v + jv);
    vb[k] = load_stream_d2<NTV>(v + jv + 2);
  }
  double xg[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int j = a0 + 4 * (threadIdx.x + k * THREADS);
    int c[4];
    c16_decode(c, d[k], rv[k], R, lane, C.ovf);
    const bool mine = j >= lo4 && j < hi;
    const int bs = __builtin_amdgcn_readlane(rv[k], 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) xg[k][q] = gather_u32(x, mine ? c[q] : bs);
  }
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const int g = threadIdx.x + k * THREADS;
    double2v p0, p1;
    p0.x = va[k].x * xg[k][0];
    p0.y = va[k].y * xg[k][1];
    p1.x = vb[k].x * xg[k][2];
    p1.y = vb[k].y * xg[k][3];
    double2v *dst = reinterpret_cast<double2v *>(lds + 4 * g);
    dst[0] = p0;
    dst[1] = p1;
  }
}

// The wavefronts at a tile's end: one step at a time, not unrolled (see stage_products_tail).
template <int THREADS, int NPT, bool NTC, bool NTV>
__device__ __forceinline__ void stage_products_c16_tail(double *__restrict__ lds, int a0, int lo4, int hi, const Col16Dev &C,
                                                        const double *__restrict__ v, const double *__restrict__ x) {
  const int lane = threadIdx.x & (kWave - 1);
  const int R = C.rec_ints;
  const int rl = lane < R ? lane : 0;
#pragma nounroll
  for (int k = 0; k < NPT / 4 - 1; ++k) {
    const int wave_j = __builtin_amdgcn_readfirstlane(a0 + 4 * ((threadIdx.x & ~(kWave - 1)) + k * THREADS));
    if (wave_j >= hi) break; // wave-uniform
    const int g = threadIdx.x + k * THREADS;
    const int j = a0 + 4 * g;
    const int rv = C.rec[static_cast<long long>(wave_j >> 8) * R + rl];
    const uint2v_a2 *p = reinterpret_cast<const uint2v_a2 *>(C.d16 + j);
    const uint2v d = NTC ? __builtin_nontemporal_load(p) : *p;
    const int jv = (j >= lo4 && j < hi) ? j : lo4;
    const double2v va = load_stream_d2<NTV>(v + jv), vb = load_stream_d2<NTV>(v + jv + 2);
    int c[4];
    c16_decode(c, d, rv, R, lane, C.ovf);
    const bool mine = j >= lo4 && j < hi;
    const int bs = __builtin_amdgcn_readlane(rv, 0);
    double xg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) xg[q] = gather_u32(x, mine ? c[q] : bs);
    double2v p0, p1;
    p0.x = va.x * xg[0];
    p0.y = va.y * xg[1];
    p1.x = vb.x * xg[2];
    p1.y = vb.y * xg[3];
    double2v *dst = reinterpret_cast<double2v *>(lds + 4 * g);
    dst[0] = p0;
    dst[1] = p1;
  }
}

template <int THREADS, int NPT, bool NTC, bool NTV>
__device__ __forceinline__ void stage_products_c16(double *__restrict__ lds, int a0, int lo4, int hi, const Col16Dev &C,
                                                   const double *__restrict__ v, const double *__restrict__ x) {
  // the wavefront's LAST step starts below hi: so do all its steps (wave-uniform)
  const int last_j = __builtin_amdgcn_readfirstlane(a0 + 4 * ((threadIdx.x & ~(kWave - 1)) + (NPT / 4 - 1) * THREADS));
  if (last_j < hi) stage_products_c16_all<THREADS, NPT, NTC, NTV>(lds, a0, lo4, hi, C, v, x);
  else stage_products_c16_tail<THREADS, NPT, NTC, NTV>(lds, a0, lo4, hi, C, v, x);
}

// Stale-plan guard of the kernels that read the encoding: they no longer read colindex, so an in-place edit of the column indices (same rowptr)
// would go unnoticed; the first wavefront of block 0 compares 64 samples of colindex with the plan's copies and raises the plan's flag
// (device_utils.hpp check_plan_guard does the same for rowptr).  64 4-byte loads per SpMV.
__device__ __forceinline__ void check_ci_guard(const int *__restrict__ ci, const Col16Dev &C, int *__restrict__ stale) {
  if (C.ci_guard != nullptr && stale != nullptr && blockIdx.x == 0 && threadIdx.x < kWave) {
    const int idx = C.guard_lo + static_cast<int>(static_cast<long long>(threadIdx.x) * C.guard_span / (kWave - 1));
    if (ci[idx] != C.ci_guard[threadIdx.x]) __hip_atomic_store(stale, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

// ---- per-row sums over a staged tile ---------------------------------------------------------------------------------------
// Every lane group (w lanes, w wave-uniform or compile-time) sums its row's span [lo, hi) of the tile.  A span of more than
// max(63, 16 w) products would keep w lanes busy for dozens of dependent LDS reads while the rest of the workgroup idles (a
// 600-non-zero row among rows of 5: ~11 us for that one lane, measured 2.5x on a circuit-like matrix), so such spans are
// handed to whole waves instead: the group leader posts (lo, hi), after a barrier the workgroup's waves take the posted spans
// round-robin (64 lanes each, DPP butterfly), and the leader picks its sum up after a second barrier.  Spans are disjoint
// pieces of one tile of at most 4096 products, so at most kTileSpans = 4096 / 64 are ever posted.
//
// Contract: called by all threads of the workgroup from uniform control flow, after the barrier that follows staging;
// sh.n must be 0 on entry (zero it once before that barrier) and is 0 again on return.  Returns the span's sum in the
// leader lane (lane == 0) when the span was posted, the lane's strided partial otherwise; the caller's later group_sum over
// the w lanes is unaffected (non-leader lanes of a posted span contribute 0).  Costs one barrier when nothing is posted.
constexpr int kTileSpans = 64; // the largest tile any caller stages is 4096 products (flat_npt 16)
template <int N> struct TileSpansN {
  int n;
  int lo[N], hi[N];
  double sum[N];
};
typedef TileSpansN<kTileSpans> TileSpans;
// (the row-block kernel's tile is 2048 products: 32 spans.  The 512 B matter: 16 KB of tile + this + the wave totals stay below 17 KB, nine
// workgroups' worth of the CU's 160 KB where the 64-span form was 20 bytes above it)
typedef TileSpansN<32> TileSpans2K;

template <int THREADS, class Spans>
__device__ __forceinline__ double tile_row_sum(const double *__restrict__ lds, Spans &sh, int lo, int hi, int lane, int w) {
  const int span = hi - lo;
  const bool posted = w < kWave && span >= 64 && span > 16 * w;
  int slot = -1;
  double s = 0.0;
  if (posted) {
    if (lane == 0) {
      slot = atomicAdd(&sh.n, 1);
      sh.lo[slot] = lo;
      sh.hi[slot] = hi;
    }
  } else {
    // four reads in flight, added in the loop's own order (bitwise the one-at-a-time sum): the loop as written is one LDS round trip per
    // addition -- s_waitcnt lgkmcnt(0) between ds_read_b64 and v_add_f64, six to eight times for a row of 22-30 products on four lanes
    int j = lo + lane;
    for (; j + 3 * w < hi; j += 4 * w) {
      const double a = lds[j], b = lds[j + w], c = lds[j + 2 * w], d = lds[j + 3 * w];
      s += a;
      s += b;
      s += c;
      s += d;
    }
    for (; j < hi; j += w) s += lds[j];
  }
  // the barrier itself tells every thread how many spans were posted (no shared variable to read after it, so a thread
  // that is late here cannot see a post from a later call)
  const int n = __syncthreads_count(posted && lane == 0);
  if (n > 0) {
    const int wave = threadIdx.x / kWave, l = threadIdx.x & (kWave - 1);
    for (int e = wave; e < n; e += THREADS / kWave) {
      double t = 0.0;
      const int b = sh.hi[e];
      for (int j = sh.lo[e] + l; j < b; j += kWave) t += lds[j];
      t = group_sum<64>(t);
      if (l == 0) sh.sum[e] = t;
    }
    __syncthreads();
    if (slot >= 0) s = sh.sum[slot];
    __syncthreads(); // every leader has its sum: the posts are consumed
    if (threadIdx.x == 0) sh.n = 0;
    __syncthreads(); // ... and the counter is back at 0 before anyone can post in a later call
  }
  return s;
}

// The same for lane groups that own R rows each (vector-row tile kernel): all R spans are summed or posted first, ONE barrier
// counts the posting lanes, and only then -- every thread being on the slow path -- is the shared post counter read.
template <int THREADS, int R, class Spans>
__device__ __forceinline__ void tile_rows_sum(const double *__restrict__ lds, Spans &sh, const int (&lo)[R], const int (&hi)[R],
                                              int lane, int w, double (&acc)[R]) {
  int slot[R];
  bool posted_any = false;
#pragma unroll
  for (int k = 0; k < R; ++k) {
    const int span = hi[k] - lo[k];
    const bool posted = w < kWave && span >= 64 && span > 16 * w;
    slot[k] = -1;
    if (posted) {
      if (lane == 0) {
        slot[k] = atomicAdd(&sh.n, 1);
        sh.lo[slot[k]] = lo[k];
        sh.hi[slot[k]] = hi[k];
        posted_any = true;
      }
    } else {
      for (int j = lo[k] + lane; j < hi[k]; j += w) acc[k] += lds[j];
    }
  }
  if (__syncthreads_count(posted_any) > 0) {
    const int n = sh.n; // all posts precede the barrier; nobody posts again before the last barrier below
    const int wave = threadIdx.x / kWave, l = threadIdx.x & (kWave - 1);
    for (int e = wave; e < n; e += THREADS / kWave) {
      double t = 0.0;
      const int b = sh.hi[e];
      for (int j = sh.lo[e] + l; j < b; j += kWave) t += lds[j];
      t = group_sum<64>(t);
      if (l == 0) sh.sum[e] = t;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < R; ++k)
      if (slot[k] >= 0) acc[k] += sh.sum[slot[k]];
    __syncthreads();
    if (threadIdx.x == 0) sh.n = 0;
    __syncthreads();
  }
}

} // namespace dev
} // namespace spmv_acc
