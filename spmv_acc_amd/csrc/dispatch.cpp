// dispatch.cpp -- one SpMV call: the strategy runners and run_spmv (split out of engine.cpp in round 4; no behaviour change).
// Reference roles: src/acc/strategy_picker.cpp:19-65 (dispatch), hip-adaptive/adaptive.cpp:16-67 (adaptive decision).
#include "engine_internal.hpp"

namespace spmv_acc {
using namespace detail;

namespace detail {

// A flat tile is one workgroup and walks its rows 256 at a time.  Where a tile owns tens of thousands of rows (hypersparse
// matrices: 50 M rows with 6000 non-zeros put all of them into ONE tile, 96 ms) the rows, not the non-zeros, need cutting: such
// matrices run the fixed row blocks instead (0.27 ms) -- the mirror image of the row-block family's rescue.
constexpr int kFlatMaxTileRows = 16384;

thread_local int t_strict_name = -1;

bool run_flat(hipStream_t st, Plan &p, double alpha, double beta, const double *x, double *y) {
  if (!ensure_flat(p, st)) return false;
  const bool strict = t_strict_name == kFlat; // (tunable strict_strategy: the caller named `flat` and means flat_tile_kernel)
  if (p.flat.max_tile_rows > kFlatMaxTileRows && tun(kT_rowblock_guard) && !strict) {
    // (guard against mutual recursion: the row-block rescue goes to row-block-plus unless rescue_flat is set, and a matrix with
    // such tiles has no row-block imbalance of the hub-row kind)
    return run_rowblock(st, p, nullptr, alpha, beta, x, y, false, 0);
  }
  p.A.cold = nullptr; // (the plan-time timings below run without gather hints)
  p.flat.col16 = nullptr;
  if (!autotune_policy(p, kFamFlat, st, [&](int pol, double *ys) { launch_flat_with(st, p, pol, 1.0, trial_beta(), x, ys); })) return false;
  if (!autotune_flat_mode(p, st, x)) return false;
  if (!autotune_flat_geometry(p, st, x)) return false;
  if (!autotune_hint(p, kFamFlat, st, [&](double *ys) { launch_flat_with(st, p, policy_for(p, kFamFlat), 1.0, trial_beta(), x, ys); })) return false;
  // the 16-bit column encoding (round 6: timed per matrix; read by the 2048-non-zero tile only, other tile sizes read colindex)
  const Col16 *c16 = nullptr;
  if (p.flat.stride == kThreads * kNnzPerThread && !flat_segment_sum() &&
      !autotune_col16(p, kFamFlat, st, [&](const Col16 *c, double *ys) {
        p.flat.col16 = c;
        launch_flat_with(st, p, policy_for(p, kFamFlat), 1.0, trial_beta(), x, ys);
      }, &c16))
    return false;
  p.flat.col16 = c16;
  // Small grids: the tile kernel's extra dependent hop (tile digest -> row extents) is not hidden by other workgroups.  Where the
  // fixed row blocks are balanced (nothing for non-zero-cut tiles to repair) the two kernels are timed once and the faster runs.
  const int rb_mode = tun(kT_flat_rowblock);
  // (a caller that pins any of the tile kernel's own choices -- cut-row form, tile size, staging order -- is asking for that kernel)
  const bool tile_pinned = tun(kT_flat_finish) >= 0 || tun(kT_flat_npt) >= 0 || tun(kT_flat_early) >= 0;
  if (rb_mode != 0 && (rb_mode > 0 || !tile_pinned) && !flat_segment_sum() && tun(kT_col16) <= 0 &&
      !t_coarse_tuning && !strict) {
    if (rb_mode > 0) {
      int vec = 1, rpb = kThreads;
      pick_rowblock_shape(p.A.m, p.A.count(), rowblock_target_for(p), &vec, &rpb);
      if (t_capturing ? (p.rowblock_ok == 1 && p.rowblock_rpb == rpb) : (probe_rowblock(p, rpb, st) && p.rowblock_ok == 1))
        return run_rowblock(st, p, nullptr, alpha, beta, x, y, false, 0);
    } else if (p.flat_rowblock_choice < 0 && !t_capturing && !by_rule()) {
      int vec = 1, rpb = kThreads;
      pick_rowblock_shape(p.A.m, p.A.count(), rowblock_target_for(p), &vec, &rpb);
      if (!probe_rowblock(p, rpb, st)) return false;
      p.flat_rowblock_choice = 0;
      if (p.rowblock_ok == 1) {
        ++t_plan_work;
        double *scratch = nullptr;
        if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
        TuneTimer timer;
        timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
        bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
        float ms2[2] = {0.f, 0.f};
        ok = ok && run_rowblock(st, p, nullptr, 1.0, trial_beta(), x, scratch, false); // (builds and tunes the row-block side)
        // (round 5: the two kernels take turns over ranking_rounds() rounds and the medians decide, with a margin of 1.5 % for the named kernel --
        // until then one block of samples each and 3 %, which left `flat` 1-3 % behind `adaptive` on stand-ins where the row blocks are faster by
        // just that: the driver's round-4 run counted flat >= 0.70 on 7 and adaptive on 6 of the same 12 matrices.  A caller who wants the tile
        // kernel whatever it costs says so: tunable strict_strategy, or flat_rowblock 0)
        ok = ok && timer.time_in_turns(st, 2, [&](int c) {
          if (c == 0) launch_flat_with(st, p, policy_for(p, kFamFlat), 1.0, trial_beta(), x, scratch);
          else (void)run_rowblock(st, p, nullptr, 1.0, trial_beta(), x, scratch, false);
        }, ranking_rounds(), ms2);
        if (!ok) return false;
        const float ms_flat = ms2[0], ms_rb = ms2[1];
        p.flat_rowblock_choice = ms_rb < 0.985f * ms_flat ? 1 : 0;
        tune_log("m %d nnz %d flat on balanced rows: tile kernel %.2f us, row blocks %.2f us -> %s", p.A.m, p.A.nnz, ms_flat * 1e3f, ms_rb * 1e3f,
                 p.flat_rowblock_choice ? "row blocks" : "tile kernel");
      }
    }
    if (rb_mode < 0 && p.flat_rowblock_choice == 1) return run_rowblock(st, p, nullptr, alpha, beta, x, y, false, 0);
  }
  launch_flat_with(st, p, policy_for(p, kFamFlat), alpha, beta, x, y);
  p.last_kernel = kKernelFlatTile;
  p.last_c16 = c16 && p.flat.stride == kThreads * kNnzPerThread && p.A.cold == nullptr ? c16->rec_ints : 0;
  return true;
}

// Opt-in (tunable `validate`): one pass over rowptr and colindex per new matrix; a matrix that fails is refused on this
// and every later call (until its plan is released) instead of sending a kernel out of bounds.
bool validate_plan(Plan &p, hipStream_t st) {
  if (p.invalid < 0) {
    if (!plan_work_allowed("validating the matrix")) return false;
    ++t_plan_work;
    int *d_flags = nullptr;
    int h = -1;
    if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d_flags), sizeof(int)), "hipMalloc validate")) return false;
    if (hip_ok(hipMemsetAsync(d_flags, 0, sizeof(int), st), "memset validate")) {
      launch_validate_csr(st, p.A, d_flags);
      if (!hip_ok(hipMemcpyAsync(&h, d_flags, sizeof(int), hipMemcpyDeviceToHost, st), "read validate") ||
          !hip_ok(hipStreamSynchronize(st), "sync validate"))
        h = -1;
    }
    (void)hipFree(d_flags);
    if (h < 0) return false;
    p.invalid = h;
  }
  if (p.invalid != 0) {
    std::string what = "matrix failed validation:";
    if (p.invalid & 1) what += " rowptr decreases or is negative;";
    if (p.invalid & 2) what += " rowptr[m] != nnz;";
    if (p.invalid & 4) what += " column index outside [0, n);";
    set_error(kErrBadArgument, what);
    return false;
  }
  return true;
}

// Opt-in (tunable guard_full): the digest of the whole rowptr, taken once per plan; then one digest + verdict pair per call, on the
// call's stream AHEAD of its SpMV kernels (the flag is up by the time the caller has synchronised and asks spmv_acc_last_error).
bool launch_full_guard(Plan &p, hipStream_t st) {
  if (!p.A.guard || !p.A.stale) return true; // (the plan runs unguarded: no slot was free)
  if (!p.have_rp_digest) {
    if (!plan_work_allowed("the full row-pointer digest (guard_full)")) return false;
    ++t_plan_work;
    // kDigestSlots sets of partial sums + one word for the build-time digest
    const size_t words = static_cast<size_t>(Plan::kDigestSlots) * kDigestMaxParts + 1;
    if (!p.d_digest_acc && !hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_digest_acc), sizeof(unsigned long long) * words), "hipMalloc digest"))
      return false;
    unsigned long long *out = p.d_digest_acc + words - 1;
    launch_rowptr_digest(st, p.A.rp, p.A.m, p.d_digest_acc);
    launch_rowptr_verdict(st, p.d_digest_acc, p.A.m, 0, nullptr, out);
    unsigned long long h = 0;
    if (!hip_ok(hipMemcpyAsync(&h, out, sizeof(h), hipMemcpyDeviceToHost, st), "read digest") || !hip_ok(hipStreamSynchronize(st), "sync digest"))
      return false;
    p.rp_digest = h;
    p.have_rp_digest = true;
  }
  unsigned long long *part = p.d_digest_acc + static_cast<size_t>(p.digest_turn++ % Plan::kDigestSlots) * kDigestMaxParts;
  launch_rowptr_digest(st, p.A.rp, p.A.m, part);
  launch_rowptr_verdict(st, part, p.A.m, p.rp_digest, p.A.stale, nullptr);
  return true;
}

// Once per matrix: would any fixed row block have to stream more than kRowblockMaxRounds tiles?
// (power-law matrices: R-MAT hub rows put millions of non-zeros into one workgroup.)
bool probe_rowblock(Plan &p, int rpb, hipStream_t st) {
  if (p.rowblock_ok >= 0 && p.rowblock_rpb == rpb) return true;
  if (!plan_work_allowed("the row-block balance probe")) return false;
  ++t_plan_work;
  p.rowblock_rpb = rpb;
  int *d_max = nullptr;
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&d_max), 2 * sizeof(int)), "hipMalloc probe")) return false;
  bool ok = hip_ok(hipMemsetAsync(d_max, 0, 2 * sizeof(int), st), "memset probe");
  if (ok) {
    const long long nblocks = (static_cast<long long>(p.A.m) + rpb - 1) / rpb;
    const long long avg_block = nblocks > 0 ? p.A.count() / nblocks : 0;
    launch_max_block_nnz(st, p.A.rp, p.A.m, rpb, static_cast<int>(avg_block), d_max);
    int h[2] = {0, 0};
    ok = hip_ok(hipMemcpyAsync(h, d_max, 2 * sizeof(int), hipMemcpyDeviceToHost, st), "read probe") &&
         hip_ok(hipStreamSynchronize(st), "sync probe");
    if (ok) {
      p.max_block_nnz = h[0];
      // balanced enough = the heaviest block needs few LDS rounds AND is not far above the average block
      // (a block that fits one tile is always fine)
      const bool few_rounds = h[0] <= kRowblockMaxRounds * kTile;
      const bool near_avg = h[0] <= kTile || h[0] <= 4 * avg_block;
      p.rowblock_ok = (few_rounds && near_avg) ? 1 : 0;
      // uneven = a fifth or more of the blocks are > 35 % away from the average block or spill into a second LDS round: fixed
      // row blocks then alternate between half-empty tiles and second rounds (striped densities), and blocks cut by
      // non-zero count do better
      p.rowblock_uneven = nblocks >= 16 && 5LL * h[1] >= nblocks;
    }
  }
  (void)hipFree(d_max);
  return ok;
}

// Row digest of the row-block family (kernels.hpp RowDigest): derived from rowptr alone, like everything else a plan holds.
bool ensure_digest(Plan &p, int rpb, hipStream_t st) {
  if (p.digest.lens && p.digest.rpb == rpb) return true;
  if (!plan_work_allowed("the row digest")) return false;
  ++t_plan_work;
  p.free_digest();
  const size_t nblocks = (static_cast<size_t>(p.A.m) + rpb - 1) / rpb;
  if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&p.digest.lens), static_cast<size_t>(p.A.m)), "hipMalloc row lengths") ||
      !hip_ok(hipMalloc(reinterpret_cast<void **>(&p.digest.base), sizeof(int) * (nblocks + 1)), "hipMalloc row-block bases")) {
    p.free_digest();
    return false;
  }
  launch_row_digest(st, p.A.rp, p.A.m, rpb, p.digest.lens, p.digest.base);
  if (!hip_ok(hipStreamSynchronize(st), "build the row digest")) { // (a later call may run on another stream)
    p.free_digest();
    return false;
  }
  p.digest.rpb = rpb;
  return true;
}

// line-enhance family with its imbalance rescue: fixed row blocks while every block stays within a
// few LDS rounds, otherwise the same tile machinery cut by non-zeros (flat) so hub rows are shared
// by many workgroups instead of serialising one.
bool run_plus(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y);

bool run_rowblock(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y,
                  bool allow_uneven_switch, int lanes_per_row) {
  int vec = 1, rpb = kThreads;
  const int forced = tun(kT_rowblock_vec);
  auto shape_for = [&](int target, int *v_out, int *r_out) {
    pick_rowblock_shape(p.A.m, p.A.count(), target, v_out, r_out);
    // (THREAD_ROW: one lane per row whatever the row length, the rows per workgroup still from the tile target)
    if (lanes_per_row > 0) *v_out = lanes_per_row;
    if (forced > 0) {
      *v_out = forced;
      *r_out = kThreads / forced;
    }
  };
  shape_for(rowblock_target_for(p), &vec, &rpb);
  if (tun(kT_rowblock_guard)) {
    if (!probe_rowblock(p, rpb, st)) return false;
    // Imbalanced (power-law) matrix: fixed row blocks would leave a few workgroups with most of the work.  The rescue
    // is the row-block-PLUS kernel -- the reference's own answer to this (hip-csr-adaptive-plus is its line-enhance
    // kernel over analysed row blocks, long rows cut into dedicated blocks) -- which measures 1 % (R-MAT scale 25),
    // 5 % (scale 22) and 17 % (scale 20) faster than the nnz-cut tiles of flat.
    if (p.rowblock_ok == 0)
      return run_plus(st, p, h_rowptr, alpha, beta, x, y);
    // Uneven but not pathological (striped densities: 60 / 20 nnz per row alternating every 300 or 5000 rows ran 196 us here and
    // 177 us in row-block-plus; 30 / 10 every 64 rows 108 vs 97 us): same answer, for the strategies that leave the choice to
    // the engine.  line / line-enhance / thread_row keep their fixed row blocks.
    if (p.rowblock_uneven && allow_uneven_switch) return run_plus(st, p, h_rowptr, alpha, beta, x, y);
  }
  const RowDigest *dg = nullptr;
  int cache_ends = 0;
  const int want_lens = tun(kT_rowlen);
  // what depends on the rows per workgroup: the row digest and the cacheable grid ends
  auto setup = [&]() {
    dg = nullptr;
    // (auto: rows of <= 8 non-zeros on average, where rowptr is >= 3.5 % of the traffic; measured at 12.6 per row the scan costs
    // more than the bytes save -- largebasis-sized 17.8 vs 17.4 us)
    if (want_lens > 0 || (want_lens < 0 && static_cast<long long>(p.A.count()) <= 8LL * p.A.m)) {
      // (inside a capture a digest that does not exist yet is simply not used: the kernel reads rowptr, same result)
      const bool have = p.digest.lens && p.digest.rpb == rpb;
      if (have || !t_capturing) {
        if (!ensure_digest(p, rpb, st)) return false;
        dg = &p.digest;
      }
    }
    // blocks at each end of the grid whose streams stay cacheable (tunable cache_ends_mb; 12 B per non-zero of stream)
    cache_ends = 0;
    if (tun(kT_cache_ends_mb) > 0 && tun(kT_zigzag) && p.A.count() > 0) {
      const long long nblocks = (static_cast<long long>(p.A.m) + rpb - 1) / rpb;
      const double bytes_per_block = 12.0 * p.A.count() / static_cast<double>(nblocks);
      cache_ends = static_cast<int>(tun(kT_cache_ends_mb) * 1048576.0 / bytes_per_block);
    }
    return true;
  };
  if (!setup()) return false;
  const int chunk = tun(kT_xcd_chunk);
  const int base_flags = chunk > 0 ? (4 | (chunk << 8)) : 0; // (bit 0 -- XCD-contiguous order --, bit 1 -- late y load -- and bit 3 -- per-lane predicated staging -- were A/B switches until round 5)
  p.A.cold = nullptr; // (the policy timing runs without gather hints)
  if (!autotune_policy(p, kFamRowblock, st, [&](int pol, double *ys) {
        const int zz = next_reverse(p) ? 64 : 0;
        launch_rowblock_stream(st, p.A, vec, rpb, base_flags | (pol << 4) | zz, 1.0, trial_beta(), x, ys, dg, cache_ends);
      }))
    return false;
  // How full a row block's 2048-product tile should be (round 6): round 3 measured 1500 products per block 1-2 % faster than 1900; with this
  // round's kernels (eight waves per SIMD, 32-bit gather offsets) 1800 is 3-6 % faster than 1500 on every stand-in of 28 and more non-zeros
  // per row and on the banded shard, 1500 stays 1.5 % ahead on the largebasis-sized one (12.6 per row), 1900 loses 3-6 % (blocks spill into a
  // second round) -- profiles/r06_rowblock_target_sweep.txt.  So the two are timed in turns, once per plan, under the policy just chosen; the
  // rule (`deterministic`, deferred tuning, captures) is 1800.
  if (tun(kT_rowblock_target) <= 0 && p.rb_target == 0 && p.stream_policy[kFamRowblock][t_beta_class] >= 0 && !t_capturing && !t_coarse_tuning &&
      !t_no_policy_timing && !by_rule()) {
    int v2[2], r2[2];
    const int cand[2] = {kRowblockTargetRule, kRowblockTargetAlt};
    for (int c = 0; c < 2; ++c) shape_for(cand[c], &v2[c], &r2[c]);
    if (v2[0] == v2[1] && r2[0] == r2[1]) {
      p.rb_target = cand[0]; // (short rows: both targets ask for more rows than a workgroup has lanes -- one shape)
    } else {
      ++t_plan_work;
      double *scratch = nullptr;
      if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
      TuneTimer timer;
      timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
      bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
      bool skip[2] = {false, false};
      // (a candidate whose blocks the balance probe refuses is out; the probe and the digest are per shape: rebuilt per turn, plan time only)
      float ms[2] = {1e30f, 1e30f};
      const int pol = policy_for(p, kFamRowblock);
      for (int c = 0; ok && c < 2; ++c) {
        vec = v2[c];
        rpb = r2[c];
        if (tun(kT_rowblock_guard)) {
          ok = probe_rowblock(p, rpb, st);
          skip[c] = ok && p.rowblock_ok == 0;
        }
      }
      for (int round = 0; ok && round < ranking_rounds(); ++round) {
        for (int c = 0; ok && c < 2; ++c) {
          if (skip[c]) continue;
          vec = v2[c];
          rpb = r2[c];
          ok = setup();
          float t = 0.f;
          ok = ok && timer.time(st, [&] {
            const int zz = next_reverse(p) ? 64 : 0;
            launch_rowblock_stream(st, p.A, vec, rpb, base_flags | (pol << 4) | zz, 1.0, trial_beta(), x, scratch, dg, cache_ends);
          }, &t);
          if (ok && t < ms[c]) ms[c] = t; // (the smaller of the rounds: the candidates alternate, each turn a median of several launches)
        }
      }
      if (!ok) return false;
      p.rb_target = (!skip[1] && (skip[0] || ms[1] < 0.985f * ms[0])) ? cand[1] : cand[0];
      tune_log("m %d nnz %d row blocks: %d products per block %.2f us, %d products %.2f us -> %d", p.A.m, p.A.nnz, cand[0], ms[0] * 1e3f, cand[1], ms[1] * 1e3f,
               p.rb_target);
    }
    shape_for(rowblock_target_for(p), &vec, &rpb);
    if (tun(kT_rowblock_guard) && !probe_rowblock(p, rpb, st)) return false;
    if (!setup()) return false;
  }
  if (!autotune_hint(p, kFamRowblock, st, [&](double *ys) {
        const int zz = next_reverse(p) ? 64 : 0;
        launch_rowblock_stream(st, p.A, vec, rpb, base_flags | (policy_for(p, kFamRowblock) << 4) | zz, 1.0, trial_beta(), x, ys, dg, cache_ends);
      }))
    return false;
  // the 16-bit column encoding (round 6): built once per plan, timed once per family against the caller's colindex
  const Col16 *c16 = nullptr;
  if (!autotune_col16(p, kFamRowblock, st, [&](const Col16 *c, double *ys) {
        const int zz = next_reverse(p) ? 64 : 0;
        launch_rowblock_stream(st, p.A, vec, rpb, base_flags | (policy_for(p, kFamRowblock) << 4) | zz, 1.0, trial_beta(), x, ys, dg, cache_ends, c);
      }, &c16))
    return false;
  const int zz = next_reverse(p) ? 64 : 0;
  launch_rowblock_stream(st, p.A, vec, rpb, base_flags | (policy_for(p, kFamRowblock) << 4) | zz, alpha, beta, x, y, dg, cache_ends, c16);
  p.last_kernel = kKernelRowblock;
  p.last_c16 = c16 && p.A.cold == nullptr ? c16->rec_ints : 0;
  return true;
}

// adaptive-plus: analysis (row blocks) + the two per-matrix timings.  MIN_NNZ_PER_BLOCK decides how full a block's
// 2048-product tile gets: 1024 (the reference's instance) half-fills it, 1920 fills it but pushes blocks with one longer
// row into a second round; which wins depends on the row-length law (FEM-like: 1536-1920, -10..-18 % time; power-law: 1024),
// so the candidates are timed once per matrix like the cache policy.
bool run_plus_prepare(Plan &p, const int *h_rowptr, hipStream_t st, const double *x) {
  auto launch = [&](int pol, double *ys) {
    launch_plus(st, p.A, p.d_pbp, p.d_pfbr, p.d_pblk, p.plus_blocks, p.plus_has_long, tun(kT_xcd_chunk), pol,
                p.d_ppartial, 1.0, trial_beta(), x, ys, next_reverse(p));
  };
  const int forced = tun(kT_plus_min_nnz);
  if (forced > 0) {
    return ensure_plus(p, h_rowptr, st, forced > 0 ? forced : kPlusMinNnz) && autotune_policy(p, kFamPlus, st, launch);
  }
  if (tun(kT_deterministic)) {
    // by rule, and a pure function of the matrix whatever was called on it before: the balance probe of the row-block shape
    // decides (hub rows: the reference's 1024, the block size that wins on power-law matrices; else 1536)
    int vec = 1, rpb = kThreads;
    pick_rowblock_shape(p.A.m, p.A.count(), rowblock_target_for(p), &vec, &rpb);
    return probe_rowblock(p, rpb, st) && ensure_plus(p, h_rowptr, st, p.rowblock_ok == 0 ? kPlusMinNnz : 1536);
  }
  if (p.plus_tuned_min > 0) return ensure_plus(p, h_rowptr, st, p.plus_tuned_min) && autotune_policy(p, kFamPlus, st, launch);
  // (inside a capture: the row blocks the plan already holds, whatever block size they were analysed with; none yet -> refused)
  if (t_capturing) return (p.plus_blocks >= 0 || ensure_plus(p, h_rowptr, st, 1536)) && autotune_policy(p, kFamPlus, st, launch);
  // (coarse: 1024 where the balance probe found hub rows -- the block size that wins on power-law matrices -- else 1536)
  if (t_coarse_tuning || defer_tuning()) // (budget spent: the coarse choice for now, the three block sizes are timed by a later call)
    return ensure_plus(p, h_rowptr, st, p.rowblock_ok == 0 ? kPlusMinNnz : 1536) && autotune_policy(p, kFamPlus, st, launch);
  // first call on this matrix: cache policy on the middle candidate, then the three block sizes under that policy
  if (!ensure_plus(p, h_rowptr, st, 1536) || !autotune_policy(p, kFamPlus, st, launch)) return false;
  double *scratch = nullptr;
  if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
  TuneTimer timer;
  timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
  bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
  const int candidates[3] = {1536, 1920, kPlusMinNnz};
  float best = 1e30f;
  int best_min = kPlusMinNnz;
  for (int c = 0; ok && c < 3; ++c) {
    ok = ensure_plus(p, h_rowptr, st, candidates[c]);
    if (!ok) break;
    float ms = 0.f;
    ok = timer.time(st, [&] { launch(policy_for(p, kFamPlus), scratch); }, &ms);
    if (ok) tune_log("m %d nnz %d beta class %d row-block-plus: MIN_NNZ_PER_BLOCK %d -> %.2f us (kept for both classes)", p.A.m, p.A.nnz, t_beta_class, candidates[c], ms * 1e3f);
    if (ok && ms < best) {
      best = ms;
      best_min = candidates[c];
    }
  }
  if (!ok) return false;
  p.plus_tuned_min = best_min;
  return ensure_plus(p, h_rowptr, st, best_min);
}

thread_local bool t_in_slab = false; // this thread is running one slab of a column-slab SpMV (no nesting)
bool ensure_segments(Plan &p, int S, hipStream_t st);
// The whole-row pass of the two-class lists (rows below slab_whole_below, each ONE run over all columns) gathers from all of x like the one-kernel
// path, and like there the plan's gather hints pay: R-MAT 24 / 25 / 26: -1.8 / -2.7 / -3.3 % of the whole SpMV (2.35 -> 2.30, 5.10 -> 4.97, 11.31 ->
// 10.93 ms).  Inside the column-slab passes they cost 10-30 % (the census' hot set is x-wide: inside a 32 MB slab almost every gather is "cold", and
// non-temporal gathers forfeit the slab's own reuse) -- profiles/r04_seg_hint_ab.txt.  Timed once per plan, this pass alone, hinted against plain;
// -1 undecided (then: hinted, the rule), 0 plain, 1 hinted.  Returns false on a HIP error only.
thread_local bool t_in_segment_timing = false;
static bool decide_whole_pass_hint(hipStream_t st, Plan &p, const double *x) {
  const int s = p.seg_slabs - 1;
  if (p.seg_rest_below <= 0 || p.seg_entries[s] == 0 || tun(kT_gather_hint) == 0 || static_cast<long long>(p.A.n) * 8 >= (1LL << 32)) return true;
  // (the census: once per plan, not inside a capture; ensure_hint keeps its own rules -- an x below 96 MB has nothing to protect)
  if (p.hint_state < 0 && !t_capturing && !ensure_hint(p, st)) return false;
  if (p.hint_state != 1 || !p.d_cold) return true;
  if (p.seg_whole_hint >= 0 || t_capturing || by_rule()) return true;
  double *scratch = nullptr;
  if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
  ++t_plan_work;
  TuneTimer timer; // (no y reset: the pass adds into whatever the scratch holds)
  float ms[2] = {0.f, 0.f};
  bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
  for (int h = 0; ok && h < 2; ++h)
    ok = timer.time(st, [&] {
      launch_segment_tiles(st, p.seg_blocks[s], 1.0, p.seg_blk[s], p.seg_row[s], p.seg_begin[s], p.seg_vptr[s], p.A.ci, p.A.v, x, p.d_seg_ys, scratch,
                           h ? p.d_cold : nullptr);
    }, &ms[h]);
  if (!ok) return false;
  p.seg_whole_hint = ms[1] < 0.985f * ms[0] ? 1 : 0;
  tune_log("m %d nnz %d: slab passes, whole-row pass: plain gathers %.1f us, hinted %.1f us -> %s", p.A.m, p.A.nnz, ms[0] * 1e3f, ms[1] * 1e3f,
           p.seg_whole_hint ? "hinted" : "plain");
  return true;
}

// the S passes over the plan's run lists (k_segment.hip); p.seg_state == 1
void run_segments(hipStream_t st, Plan &p, double alpha, double beta, const double *x, double *y) {
  p.last_kernel = kKernelSlabPasses;
  const int whole = p.seg_rest_below > 0 ? p.seg_slabs - 1 : -1; // the whole-row pass of the two-class lists
  // (a call that is itself a trial launch of a timing phase decides nothing: time_against_segments has asked before it started its clock)
  if (!t_in_segment_timing && !decide_whole_pass_hint(st, p, x)) return;
  const bool whole_hinted = whole >= 0 && p.hint_state == 1 && p.d_cold && p.seg_whole_hint != 0 && tun(kT_gather_hint) != 0 &&
                            static_cast<long long>(p.A.n) * 8 < (1LL << 32);
  const unsigned char *whole_cold = whole_hinted ? p.d_cold : nullptr;
  launch_guard_check(st, p.A); // (the passes read run lists, not rowptr: the caller's rowptr is checked here)
  if (beta != 1.0 || p.A.yin) launch_scale_y(st, p.A.m, beta, y, p.A.yin);
  for (int s = 0; s < p.seg_slabs; ++s) {
    if (p.seg_entries[s] == 0) continue;
    launch_segment_tiles(st, p.seg_blocks[s], alpha, p.seg_blk[s], p.seg_row[s], p.seg_begin[s], p.seg_vptr[s], p.A.ci, p.A.v, x, p.d_seg_ys, y,
                         s == whole ? whole_cold : nullptr);
    if (p.seg_pieces[s]) launch_segment_merge(st, p.seg_pieces[s], p.seg_cut[s], p.seg_entries[s], p.seg_row[s], p.d_seg_ys, y);
  }
}
// automatic mode: slabs of about 32 MB of x (R-MAT scale 25, x = 256 MB, S = 4 / 8 / 12 / 16: 5.92 / 5.31 / 5.59 / 6.08 ms, 7.27 without;
// scale 24, x = 128 MB, S = 4 / 8: 2.37 / 2.59 ms, 3.21 without)
int seg_auto_slabs(int n) {
  const long long per = static_cast<long long>(tun(kT_slab_kb) > 0 ? tun(kT_slab_kb) : 32768) << 10; // (tunable slab_kb: tests reach S = 16 at test size)
  const long long s = (static_cast<long long>(n) * 8 + per / 2) / per;
  return s < 2 ? 2 : (s > 16 ? 16 : static_cast<int>(s));
}

bool run_plus(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y) {
  p.A.cold = nullptr; // (the prepare timings run without hints)
  auto launch_here = [&](double a, double b, double *yy) {
    launch_plus(st, p.A, p.d_pbp, p.d_pfbr, p.d_pblk, p.plus_blocks, p.plus_has_long, tun(kT_xcd_chunk), policy_for(p, kFamPlus),
                p.d_ppartial, a, b, x, yy, next_reverse(p));
  };
  // Builds the run lists (once) and times the slab passes against this kernel as it stands; ms[0] row-block-plus, ms[1] the passes.
  // Returns false on an error; *timed says whether both timings exist (no room for the lists / rows not ordered: they do not).
  auto time_against_segments = [&](float ms[2], bool *timed) {
    *timed = false;
    // the lists are an optimisation: a matrix that leaves no room for them (or for the build's S x (m + 1) temporaries) keeps the
    // one-kernel path instead of failing the SpMV
    const int S_auto = seg_auto_slabs(p.A.n);
    if (p.seg_state < 0) {
      size_t free_b = 0, total_b = 0;
      const size_t build_bytes = (2 * static_cast<size_t>(S_auto) + 4) * (static_cast<size_t>(p.A.m) + 1) * sizeof(int) + (static_cast<size_t>(p.A.count()) / 4) * 12;
      const bool room = hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > 2 * build_bytes;
      (void)hipGetLastError();
      if (room && last_error_code_only() == kOk && !ensure_segments(p, S_auto, st)) {
        (void)hipGetLastError();
        tune_log("m %d nnz %d: slab_segments: the run lists could not be built (%s), row-block-plus stays", p.A.m, p.A.nnz, last_error_string());
        clear_error();
        p.free_segments();
      }
    }
    if (p.seg_state != 1) return true;
    ++t_plan_work;
    double *scratch = nullptr;
    if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
    TuneTimer timer;
    timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
    bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
    const double *keep_yin = p.A.yin;
    p.A.yin = nullptr; // (the trial runs update the scratch vector in place)
    ok = ok && decide_whole_pass_hint(st, p, x); // (the passes' own choice first, outside the clock of the comparison)
    t_in_segment_timing = true;
    // (launches of >= 4 ms are sampled once each; a decision that is kept for the life of the plan gets a second sample of both -- the smaller
    // counts -- only where the first ones are within a third of each other: on R-MAT 25, 8.5 ms against 5.0, the second pair was 13 ms of a
    // 97 ms first call and could not have changed anything)
    ok = ok && timer.time(st, [&] { launch_here(1.0, trial_beta(), scratch); }, &ms[0]) &&
         timer.time(st, [&] { run_segments(st, p, 1.0, trial_beta(), x, scratch); }, &ms[1]);
    if (ok && ms[0] < 1.33f * ms[1] && ms[1] < 1.33f * ms[0]) {
      float again[2] = {0.f, 0.f};
      ok = timer.time(st, [&] { launch_here(1.0, trial_beta(), scratch); }, &again[0]) &&
           timer.time(st, [&] { run_segments(st, p, 1.0, trial_beta(), x, scratch); }, &again[1]);
      if (ok) ms[0] = std::min(ms[0], again[0]), ms[1] = std::min(ms[1], again[1]);
    }
    t_in_segment_timing = false;
    p.A.yin = keep_yin;
    *timed = ok;
    return ok;
  };
  // (tunable strict_strategy: `line_enhance` / `line` named by the caller keep to the row-block(-plus) kernel)
  const bool slabs_auto = tun(kT_slab_segments) < 0 && !t_in_slab && t_strict_name != kLineEnhance && t_strict_name != kLine;
  // Power-law columns (the column census finds a hot set, x far beyond the L2s): this kernel is bound by gathers that miss, and the slab
  // passes over run lists (k_segment.hip) usually replace it.  So the passes are decided FIRST, against this kernel in its COARSE
  // configuration -- the rule's cache policy and block size, no hints: nothing timed for it -- and the kernel's own choices (three block
  // sizes, three cache policies, hints: eight trial launches of 7-8 ms each on R-MAT 25, 60 of the 145 ms its first call took) are only
  // timed when the passes do not win clearly.  Clearly = by 15 %: tuned and hinted, this kernel gains up to ~12 % on its coarse form.
  if (slabs_auto && p.seg_choice < 0 && !p.seg_early_tried && !t_capturing && !by_rule() && tun(kT_gather_hint) != 0) {
    if (!ensure_hint(p, st)) return false;
    p.seg_early_tried = true; // (once per plan: an undecided outcome leaves the question to the comparison against the TUNED kernel below)
    if (p.hint_state == 1) {
      const bool was_coarse = t_coarse_tuning;
      t_coarse_tuning = t_no_policy_timing = true;
      const bool prepared = run_plus_prepare(p, h_rowptr, st, x);
      t_coarse_tuning = was_coarse;
      t_no_policy_timing = false;
      if (!prepared) return false;
      float ms[2] = {0.f, 0.f};
      bool timed = false;
      if (!time_against_segments(ms, &timed)) return false;
      if (timed && ms[1] < 0.85f * ms[0]) p.seg_choice = 1;
      if (timed)
        tune_log("m %d nnz %d beta class %d: row-block-plus (coarse) %.2f us, %d column-slab passes over run lists %.2f us -> %s", p.A.m, p.A.nnz, t_beta_class,
                 ms[0] * 1e3f, seg_auto_slabs(p.A.n), ms[1] * 1e3f, p.seg_choice == 1 ? "slab passes" : "not decided: the kernel is tuned first");
      if (!timed) p.seg_choice = 0;
    }
  }
  if (slabs_auto && p.seg_choice == 1) {
    // (a plan that adopted the choice from the tune cache builds its lists here; inside a capture only lists that exist are used)
    if (p.seg_state != 1 && !t_capturing && last_error_code_only() == kOk && !ensure_segments(p, seg_auto_slabs(p.A.n), st)) {
      (void)hipGetLastError(); // (no room for the lists this time: the one-kernel path)
      clear_error();
      p.free_segments();
      p.seg_choice = 0;
    }
    if (p.seg_state == 1) {
      run_segments(st, p, alpha, beta, x, y);
      return true;
    }
  }
  if (!run_plus_prepare(p, h_rowptr, st, x)) return false;
  if (!autotune_hint(p, kFamPlus, st, [&](double *ys) { launch_here(1.0, trial_beta(), ys); })) return false;
  // the passes were not clearly faster than the coarse kernel: once more against the tuned one, the faster (by 5 %) stays
  if (slabs_auto && p.hint_state == 1 && p.seg_choice < 0 && !t_capturing && !by_rule()) {
    // (also while adaptive is timing its kernel families: row-block-plus is then timed as what it will run -- on R-MAT 25 the one-kernel
    // path beats flat by 0.5 % only, 7.26 against 7.29 ms, and a family choice made on that would never meet the 5.3 ms of the passes)
    float ms[2] = {0.f, 0.f};
    bool timed = false;
    if (!time_against_segments(ms, &timed)) return false;
    p.seg_choice = timed && ms[1] < 0.95f * ms[0] ? 1 : 0;
    if (timed)
      tune_log("m %d nnz %d beta class %d: row-block-plus %.2f us, %d column-slab passes over run lists %.2f us -> %s", p.A.m, p.A.nnz,
               t_beta_class, ms[0] * 1e3f, seg_auto_slabs(p.A.n), ms[1] * 1e3f, p.seg_choice ? "slab passes" : "row-block-plus");
    if (p.seg_choice == 0) p.free_segments(); // (the lists of a matrix that does not use them: 12 B per run back)
    if (p.seg_choice == 1 && p.seg_state == 1) {
      run_segments(st, p, alpha, beta, x, y);
      return true;
    }
  }
  launch_here(alpha, beta, y);
  p.last_kernel = kKernelPlus;
  return true;
}

// adaptive, measured: the reference decides from four rowptr samples which kernel family a matrix gets (adaptive.cpp:16-67).
// Which family wins depends on more than those samples say -- fixed row blocks on evenly filled matrices, blocks cut by
// non-zero count where the density varies (quarters, stripes), non-zero-cut tiles where many rows are hundreds to thousands
// long (lognormal row lengths with sigma >= 1: flat 116 us, row blocks 127-130 us; 2000 rows of 3000 nnz in an FEM-like matrix:
// 125 vs 149 us, tools/rowlaw_bench.py) -- so the three are timed once per matrix, each after its own first call has built
// and tuned its plan, and the fastest is kept.  The sample-based rules remain as the untimed form (tunable adaptive_timed 0).
bool run_adaptive_timed(hipStream_t st, Plan &p, const int *h_rowptr, double alpha, double beta, const double *x, double *y) {
  auto run_family = [&](int f, double a, double b, double *yy) {
    switch (f) {
    case 0: return run_rowblock(st, p, h_rowptr, a, b, x, yy);
    case 1: return run_plus(st, p, h_rowptr, a, b, x, yy);
    default: return run_flat(st, p, a, b, x, yy);
    }
  };
  const int cls = t_beta_class;
  if (p.adaptive_family[cls] < 0 && t_capturing) {
    // no timing inside a capture: the family the other beta class settled on (prepared matrices: spmv_acc_prepare times beta = 1),
    // whose plan exists; with neither class timed the call is refused
    if (p.adaptive_family[cls ^ 1] >= 0) return run_family(p.adaptive_family[cls ^ 1], alpha, beta, y);
    return plan_work_allowed("adaptive: timing the kernel families on this matrix");
  }
  // The comparison has two parts: a FIRST LOOK (each family built with its default sub-choices and timed once) and a SECOND LOOK at every
  // family within 8 % of the fastest.  Under the call's tuning budget (defer_tuning) the second look may fall to a later call: the first
  // look's choice serves until then (adaptive_provisional) and its timings are kept in the plan.
  auto decide = [&](const float *ms, bool ranked) {
    // fixed row blocks unless another family is at least 3 % faster (short kernels time within ~2 %); 1.5 % once the families have been ranked in
    // turns (the second look, round 5)
    const float margin = ranked ? 0.985f : 0.97f;
    int best_family = ms[0] < 1e29f ? 0 : 1;
    for (int f = 1; f < 3; ++f)
      if (ms[f] < (best_family == 0 ? margin * ms[0] : ms[best_family])) best_family = f;
    return best_family;
  };
  // The first look is incremental under the budget: a family that has not been timed yet is built and timed only while the call may still
  // spend (the first call times fixed row blocks at least; the others follow, one per later call if need be), and until all three are in, the
  // best of the measured ones serves.
  float *ms = p.adaptive_ms[cls];
  const bool open = p.adaptive_family[cls] < 0 || p.adaptive_provisional[cls];
  if (open && !t_capturing && !(p.adaptive_family[cls] >= 0 && defer_tuning())) {
    ++t_plan_work;
    double *scratch = nullptr;
    if (!(scratch = tune_scratch(static_cast<size_t>(p.A.m)))) return false;
    TuneTimer timer;
    timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
    bool ok = timer.ok && hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
    // The families are compared in the caller's beta class: with beta != 0 every row also reads its old y, which is a large
    // share of the traffic where rows hold one or two non-zeros and ranks the families differently (15 M rows of ~1 nnz:
    // flat looked 3 % faster than the row blocks at beta = 0 and is 9 % slower at beta = 1).
    const double beta_trial = trial_beta();
    t_coarse_tuning = true;
    bool any_measured = false;
    int timed_here = 0;
    for (int f = 0; f < 3; ++f) any_measured = any_measured || ms[f] < 1e29f;
    for (int f = 0; ok && f < 3; ++f) {
      if (ms[f] < 1e29f || p.adaptive_skipped[cls][f]) continue;
      // At least one family is timed whatever the budget -- the first call needs a kernel, and every later call that gets here must make
      // progress.  A further one is only started while most of the budget is left: building a family's plan is structural work of unknown
      // size (the row-block analysis, its tables and their allocations took 4 ms on the headline matrix, 25 SpMVs' worth).
      const bool must_progress = timed_here == 0 && (!any_measured || p.calls > 1);
      if (!must_progress && (defer_tuning() || budget_spent_fraction() > 0.25)) {
        t_tuning_deferred = true;
        break;
      }
      ok = run_family(f, 1.0, beta_trial, scratch); // builds this family's plan (sub-choices at their defaults)
      if (!ok) break;
      if (f == 0 && p.rowblock_ok == 0) { // fixed row blocks were rescued: that run WAS family 1
        p.adaptive_skipped[cls][0] = true;
        continue;
      }
      ok = timer.time(st, [&] { (void)run_family(f, 1.0, beta_trial, scratch); }, &ms[f]);
      any_measured = any_measured || ok;
      ++timed_here;
    }
    bool unmeasured = false;
    for (int f = 0; f < 3; ++f) unmeasured = unmeasured || (ms[f] > 1e29f && !p.adaptive_skipped[cls][f]);
    // second look at every family within 8 % of the fastest: the choice is kept for the life of the plan, and two families
    // 3 % apart changed places from process to process on single timings (the headline matrix ran fixed row blocks in one
    // run and row-block-plus in the next); the smaller of the two timings counts
    bool looked_twice = false;
    if (ok && !unmeasured && (timed_here == 0 || !defer_tuning())) { // (a call that timed no family takes the second look whatever its budget: progress)
      float fastest = ms[0];
      for (int f = 1; f < 3; ++f) fastest = ms[f] < fastest ? ms[f] : fastest;
      // (round 5: the families within 8 % take turns over ranking_rounds() rounds and their medians replace the first look's figures -- until then one
      // more block of samples per family, the smaller of the two counting, which ranked near-equal families by the moment they were timed)
      bool skip[3];
      int close = 0;
      for (int f = 0; f < 3; ++f) close += (skip[f] = ms[f] > 1.08f * fastest) ? 0 : 1;
      if (close > 1) {
        float again[3] = {1e30f, 1e30f, 1e30f};
        ok = timer.time_in_turns(st, 3, [&](int f) { (void)run_family(f, 1.0, beta_trial, scratch); }, ranking_rounds(), again, skip);
        for (int f = 0; ok && f < 3; ++f)
          if (!skip[f]) ms[f] = ranking_rounds() > 1 ? again[f] : (again[f] < ms[f] ? again[f] : ms[f]);
      }
      looked_twice = ok;
    }
    t_coarse_tuning = false;
    const int best_family = decide(ms, looked_twice && ranking_rounds() > 1);
    tune_log("m %d nnz %d adaptive (beta %s 0): fixed row blocks %.2f us, row-block-plus %.2f us, flat %.2f us -> family %d%s", p.A.m, p.A.nnz,
             beta != 0.0 ? "!=" : "==", ms[0] * 1e3f, ms[1] * 1e3f, ms[2] * 1e3f, best_family,
             looked_twice ? "" : (unmeasured ? " (so far: the other families wait for a later call's tuning budget)" : " (first look; the second look waits for a later call's tuning budget)"));
    if (!ok) return false;
    p.adaptive_family[cls] = best_family;
    p.adaptive_provisional[cls] = !looked_twice;
    if (!looked_twice) t_tuning_deferred = true;
  }
  return run_family(p.adaptive_family[cls], alpha, beta, y);
}

} // namespace detail


// The one plan-resident copy of VALUES is the column slabs' (opt-in).  A caller that changes values in place -- which every other
// plan survives -- refreshes it with this: one scatter pass over the matrix, the slabs' structure (and their plans) stay.
int refresh_values(const int *d_rowptr) {
  std::vector<std::shared_ptr<Plan>> todo;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_plans)
      if (std::get<1>(kv.first) == d_rowptr && kv.second->d_slab_rp) todo.push_back(kv.second);
  }
  hipStream_t st = t_stream;
  for (auto &p : todo) {
    std::lock_guard<std::mutex> plan_lock(p->mu);
    if (!p->d_slab_rp) continue;
    launch_slab_scatter(st, p->A, p->slab_width, p->slab_count, p->d_slab_rp, p->d_slab_off, p->d_slab_ci, p->d_slab_v, /*values_only=*/true);
  }
  return static_cast<int>(todo.size());
}

UnboundedTuningScope::UnboundedTuningScope() { ++t_unbounded_tuning; }
UnboundedTuningScope::~UnboundedTuningScope() { --t_unbounded_tuning; }
FlatSegmentSumScope::FlatSegmentSumScope() : prev(t_flat_segment_sum) { t_flat_segment_sum = true; }
FlatSegmentSumScope::~FlatSegmentSumScope() { t_flat_segment_sum = prev; }

// Column-slab blocking over the plan's slab-major COPY (k_slab.hip): S consecutive SpMVs of this strategy on the plan's slabs, the first one applying
// beta (and reading y_in), the others accumulating into y.  Each slab is an ordinary matrix with a plan of its own.
static bool run_col_slabs(Plan &p, int S, int strategy, hipStream_t st, double alpha, double beta, int m, int n, const double *dx, double *dy) {
  if (!ensure_slabs(p, S, st)) return false;
  launch_guard_check(st, p.A); // (the slabs' kernels check the slabs: the caller's rowptr is checked here)
  // y = beta * y_in first (nothing to do for beta == 1 in place), then every slab: y_s = alpha * A_s x over the slab's non-empty
  // rows (an ordinary SpMV of a smaller matrix, beta = 0) and y[rowid] += y_s
  if (beta != 1.0 || p.A.yin) launch_scale_y(st, m, beta, dy, p.A.yin);
  const bool outer = !t_in_slab;
  t_in_slab = true;
  for (int s = 0; s < S && last_error_code_only() == kOk; ++s) {
    const long long o = p.slab_off[s];
    const int ms = p.slab_rows[s];
    if (ms == 0) continue; // an empty slab adds nothing
    run_spmv(strategy, 0, alpha, 0.0, ms, n, static_cast<int>(p.slab_off[s + 1] - o), nullptr, p.slab_crp[s], p.d_slab_ci + o, p.d_slab_v + o, dx,
             p.d_slab_ys, nullptr);
    if (last_error_code_only() == kOk) launch_slab_merge(st, ms, p.slab_rowid[s], p.d_slab_ys, dy);
  }
  if (outer) t_in_slab = false;
  p.last_kernel = kKernelColSlabs;
  return last_error_code_only() == kOk;
}

// The AUTOMATIC slab-major copy (round 6; tunable col_slabs = -1, the default).  Round 3 built the copy form and left it opt-in because the plan
// then holds 12 B per non-zero -- and VALUES; on R-MAT 25 it is 12 % faster than the run-list passes (4.4 against 5.0 ms: every 128-B line of the
// streams is used whole instead of being fetched by every pass that owns a part of it).  Rule: a plan whose own timed choice is the slab passes
// (power-law columns, x far beyond the L2s), not under `deterministic` / `strict_strategy`, not inside a capture, once it has served
// kSlabCopyAfterCalls calls (or inside spmv_acc_prepare: the caller pays up front), on a device with free memory >= 3 x 12 B per non-zero, builds
// the copy, lets the slabs' plans settle, times copy against passes in turns and keeps the faster by >= 3 %.
// The values: before EVERY use of the copy kValueSamples evenly spaced samples of the caller's values are compared with what they were when the
// copy was made (one small kernel + a stream synchronisation: these calls block, which is why captures keep the passes); on a difference the
// copy's values are scattered again (one pass over the matrix) and the samples retaken, then the call proceeds with the new values.
// Returns the slab count the copy should serve THIS call with, 0 = the ordinary path.
static int slab_copy_auto(Plan &p, int strategy, hipStream_t st, int n, const double *dx) {
  const bool hook = tun(kT_col_slabs) == -2; // (tests: the copy replaces the passes wherever the plan holds run lists, forced or chosen, whatever the timing says)
  if (p.slab_copy_choice == 0 || p.seg_state != 1 || !(hook || (p.seg_choice == 1 && tun(kT_slab_segments) < 0)) || tun(kT_deterministic) || t_strict_name >= 0 ||
      t_capturing || !dx)
    return 0;
  const int S = p.seg_slabs - (p.seg_rest_below > 0 ? 1 : 0); // the column slabs of the passes this plan already runs
  if (S < 2 || S > 16) return 0;
  const bool adopted = p.slab_copy_choice == 1 && !p.d_slab_rp; // (the tune cache says the copy won on this matrix in an earlier process: built here, not timed again)
  if (p.slab_copy_choice < 0 || adopted) {
    if (!(t_unbounded_tuning > 0 || (p.calls > static_cast<unsigned long long>(kSlabCopyAfterCalls) && !defer_tuning()))) return 0;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 36ull * static_cast<size_t>(p.A.count())) {
      (void)hipGetLastError();
      p.slab_copy_choice = 0;
      tune_log("m %d nnz %d: slab-major copy not built: %.1f GB free, 3 x 12 B per non-zero = %.1f GB wanted", p.A.m, p.A.nnz, free_b / 1e9, 36e-9 * p.A.count());
      return 0;
    }
    ++t_plan_work;
    const int count = p.A.count() < kValueSamples ? p.A.count() : kValueSamples;
    bool ok = ensure_slabs(p, S, st) &&
              hip_ok(hipMalloc(reinterpret_cast<void **>(&p.d_value_samples), 12 * static_cast<size_t>(count) + 8), "hipMalloc value samples") &&
              hip_ok(hipHostMalloc(reinterpret_cast<void **>(&p.h_values_changed), sizeof(int)), "hipHostMalloc value flag");
    double *scratch = ok ? tune_scratch(static_cast<size_t>(p.A.m)) : nullptr;
    ok = ok && scratch != nullptr;
    if (ok) {
      *p.h_values_changed = 0;
      p.value_samples = count;
      launch_value_samples(st, p.A.v, p.A.ci, p.A.nnz0, p.A.count() - 1, count, p.d_value_samples, nullptr);
      ok = hip_ok(hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(p.A.m), st), "memset tune y");
      // the slabs' own plans settle first (every slab is an ordinary matrix with its own timed choices), outside any budget
      struct Unbounded {
        Unbounded() { ++t_unbounded_tuning; }
        ~Unbounded() { --t_unbounded_tuning; }
      } unbounded;
      const double saved_budget = t_budget_spmvs;
      t_budget_spmvs = 0.0;
      for (int round = 0; ok && round < 4 && !adopted; ++round) {
        const unsigned w0 = t_plan_work;
        ok = run_col_slabs(p, S, strategy, st, 1.0, trial_beta(), p.A.m, n, dx, scratch) && hip_ok(hipStreamSynchronize(st), "settle the slabs' plans");
        if (t_plan_work == w0) break;
      }
      TuneTimer timer;
      timer.set_reset(scratch, sizeof(double) * static_cast<size_t>(p.A.m));
      float ms[2] = {0.f, 0.f};
      t_in_segment_timing = true;
      ok = ok && timer.ok && (adopted || timer.time_in_turns(st, 2, [&](int c) {
        if (c == 0) run_segments(st, p, 1.0, trial_beta(), dx, scratch);
        else (void)run_col_slabs(p, S, strategy, st, 1.0, trial_beta(), p.A.m, n, dx, scratch);
      }, 2, ms));
      t_in_segment_timing = false;
      t_budget_spmvs = saved_budget;
      if (ok && !adopted) {
        p.slab_copy_choice = (ms[1] < 0.97f * ms[0] || tun(kT_col_slabs) == -2) ? 1 : 0; // (-2: tests keep the copy whatever the timing says)
        tune_log("m %d nnz %d: slab passes over run lists %.1f us, slab-major copy (%d slabs, %.2f GB held) %.1f us -> %s", p.A.m, p.A.nnz, ms[0] * 1e3f, S,
                 12e-9 * p.A.count(), ms[1] * 1e3f, p.slab_copy_choice ? "the copy (values guarded by samples)" : "the passes");
      }
    }
    if (!ok || p.slab_copy_choice != 1) {
      if (!ok) {
        (void)hipGetLastError();
        clear_error(); // (no room after all, or a failed launch of a trial: the copy is an optimisation, the passes serve)
      }
      p.free_slabs();
      p.slab_copy_choice = 0;
      return 0;
    }
  }
  // the caller's values against the copy's, before the copy is used
  *p.h_values_changed = 0;
  launch_value_samples(st, p.A.v, p.A.ci, p.A.nnz0, p.A.count() - 1, p.value_samples, p.d_value_samples, p.h_values_changed);
  if (!hip_ok(hipStreamSynchronize(st), "compare the value samples")) return 0;
  if (*p.h_values_changed & 2) {
    // column indices edited in place: the copy's structure is stale.  It goes; the passes -- which read the caller's arrays, and for which a moved
    // column only costs locality (k_segment.hip) -- serve from here on.
    tune_log("m %d nnz %d: the caller's column indices changed under the slab-major copy: the copy is dropped, the run-list passes serve", p.A.m, p.A.nnz);
    p.free_slabs();
    p.slab_copy_choice = 0;
    return 0;
  }
  if (*p.h_values_changed) {
    launch_slab_scatter(st, p.A, p.slab_width, p.slab_count, p.d_slab_rp, p.d_slab_off, p.d_slab_ci, p.d_slab_v, /*values_only=*/true);
    launch_value_samples(st, p.A.v, p.A.ci, p.A.nnz0, p.A.count() - 1, p.value_samples, p.d_value_samples, nullptr);
    ++p.values_refreshed;
    tune_log("m %d nnz %d: the caller's values changed under the slab-major copy: refreshed (%u so far)", p.A.m, p.A.nnz, p.values_refreshed);
  }
  return S;
}

namespace {
thread_local bool t_early_clock = false; // run_spmv's rule-twin path: the tuning pass's budget clock started when the CALL began (twin's serve included)
thread_local std::chrono::steady_clock::time_point t_early_began;

void run_spmv_call(int strategy, int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
                   const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx, double *dy,
                   const double *dy_in);
} // namespace

// One SpMV call.  Round 6 (review item 7): while a matrix's per-matrix timings are open, its calls are SERVED BY RULE -- by the plan's rule twin,
// a second plan of the same arrays that decides everything the way `deterministic = 1` does -- and the timings advance beside them against a scratch
// y, under the same budget (first_call_budget / later_call_budget, the clock started when the call began, the twin's work included).  So the
// iterations before a plan settles are bitwise equal to each other (rounds 2-5 served them with the choices as far as they had come: the result's
// last bits could change from one call to the next while the plan settled), and from the first settled call on the timed choices serve, bitwise
// equal among themselves.  One switch per (matrix, strategy, beta class) -- Plan::settled_for, never taken back: what another strategy or the other
// beta class still has open does not send a settled kind of call back to the twin -- at a call the caller can see (spmv_acc_query_plan's `settled`).
// Outside this: `deterministic = 1` (always the rule), `deterministic = -1` (the earlier behaviour), spmv_acc_prepare (settles before the first
// call), stream captures (time nothing), the slabs of a slab-major copy (derived matrices of a settled parent).
void run_spmv(int strategy, int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
              const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx, double *dy,
              const double *dy_in) {
  apply_env_tunables();
  // (a forced slab-major copy -- col_slabs >= 2 -- is pinned by the caller and holds the matrix a second time: no twin of that)
  bool early = rule_until_settled() && !t_in_slab && !t_rule_twin && t_unbounded_tuning == 0 && m > 0 && d_rowptr && dy && strategy >= 0 &&
               strategy < kStrategyCount && strategy < 32 && tun(kT_col_slabs) < 2;
  bool capturing = false;
  if (early) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    capturing = hipStreamIsCapturing(t_stream, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
  }
  if (early) early = !plan_settled_for(d_rowptr, d_colindex, d_value, m, n, strategy, beta != 0.0 ? 1 : 0);
  if (!early) {
    run_spmv_call(strategy, trans, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy, dy_in);
    return;
  }
  if (capturing && !rule_twin_exists(d_rowptr, d_colindex, d_value, m, n)) {
    // (a capture on a plan that spmv_acc_prepare settled for the other beta class, or that no eager call of this kind has met: no twin to record --
    // the plan's own kernels with what it holds, as before round 6)
    run_spmv_call(strategy, trans, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy, dy_in);
    return;
  }
  if (capturing) {
    // a capture of an unsettled plan records what serves such a plan: the twin's kernels (nothing is timed inside a capture) -- and the twin stays
    // alive for the graph's sake when the plan settles later
    struct TwinScope {
      TwinScope() { t_rule_twin = true; }
      ~TwinScope() { t_rule_twin = false; }
    } twin;
    run_spmv_call(strategy, trans, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy, dy_in);
    return;
  }
  const auto began = std::chrono::steady_clock::now();
  const int err_before = last_error_code_only();
  // 1. the caller's y, by rule
  int served_kernel = -1, served_c16 = 0;
  {
    struct TwinScope {
      TwinScope() { t_rule_twin = true; }
      ~TwinScope() { t_rule_twin = false; }
    } twin;
    run_spmv_call(strategy, trans, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, dy, dy_in);
    if (const std::shared_ptr<Plan> tp = t_last_plan.lock()) {
      served_kernel = tp->last_kernel;
      served_c16 = tp->last_c16;
    }
  }
  if (last_error_code_only() != err_before && last_error_code_only() != kOk && last_error_code_only() != kErrUnsupportedTrans) return; // (the call failed: nothing to tune)
  const double twin_prepare_us = t_last_prepare_us; // (spmv_acc_last_prepare_us reports the call's whole preparation: the twin's and the plan's)
  // 2. the timings, one budget's worth, against a scratch y (tune_scratch: this thread's, released when the pass returns)
  double *scratch = tune_scratch(static_cast<size_t>(m));
  if (!scratch) {
    clear_error(); // (no room for a scratch y: the plan stays open, the twin keeps serving)
    return;
  }
  (void)hipMemsetAsync(scratch, 0, sizeof(double) * static_cast<size_t>(m), t_stream);
  t_early_began = began;
  t_early_clock = true;
  run_spmv_call(strategy, 0, alpha, beta, m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value, dx, scratch, nullptr);
  t_early_clock = false;
  t_last_prepare_us += twin_prepare_us;
  // 3. what the caller can ask about is the plan being settled, served so far by its twin's kernel
  if (const std::shared_ptr<Plan> p = t_last_plan.lock()) {
    std::lock_guard<std::mutex> plan_lock(p->mu);
    const int cls = beta != 0.0 ? 1 : 0;
    if (!((p->settled_for[cls] >> strategy) & 1u)) {
      p->last_kernel = served_kernel;
      p->last_c16 = served_c16;
    }
  }
  if (plan_settled_for(d_rowptr, d_colindex, d_value, m, n, strategy, beta != 0.0 ? 1 : 0)) drop_rule_twin(d_rowptr, d_colindex, d_value, m, n);
}

namespace {
void run_spmv_call(int strategy, int trans, double alpha, double beta, int m, int n, int nnz, const int *h_rowptr,
                   const int *d_rowptr, const int *d_colindex, const double *d_value, const double *dx, double *dy,
                   const double *dy_in) {
  const int strategy_named = strategy;
  if (trans != 0) {
    // the reference never reads `trans` (only operation_none is supported, api/spmv.h:13); it computes
    // the non-transposed product.  Same here, but the mismatch is reported out of band.
    set_error(kErrUnsupportedTrans, "only operation_none is supported; computed y = alpha*A*x + beta*y");
  }
  apply_env_tunables();
  if (m <= 0) return;
  if (m > INT_MAX - (1 << 16)) { // row arithmetic in the kernels is int32 with a workgroup's worth of slack, like nnz
    set_error(kErrTooLarge, "row count does not leave room for block arithmetic in int32; shard the matrix");
    return;
  }
  if (!d_rowptr || !dy || (n > 0 && !dx)) {
    set_error(kErrBadArgument, "null rowptr / x / y");
    return;
  }
  if (dy_in == dy) dy_in = nullptr; // in place after all
  if (dy_in && beta != 0.0) {
    // out of place: every row reads y_in[row] and writes y_out[row] exactly once, so the two vectors may be anything but
    // PARTLY overlapping (a shifted view of the same buffer would let one row's store land on another row's unread old value)
    const uintptr_t a = reinterpret_cast<uintptr_t>(dy_in), b = reinterpret_cast<uintptr_t>(dy), bytes = sizeof(double) * static_cast<uintptr_t>(m);
    if (a < b + bytes && b < a + bytes) {
      set_error(kErrBadArgument, "y_in and y_out overlap without being the same vector");
      return;
    }
  }
  hipStream_t st = t_stream;
  note_stream_use();
  {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    t_capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
  }
  if (strategy < 0 || strategy >= kStrategyCount) {
    set_error(kErrUnknownStrategy, "unknown strategy id");
    return;
  }
  const std::shared_ptr<Plan> p = get_plan(m, n, nnz, h_rowptr, d_rowptr, d_colindex, d_value);
  if (!p) return;
  t_last_plan = p;
  // the timing phases of this call share one scratch y (tune_scratch); it goes when the outermost call returns
  struct ScratchScope {
    bool outer;
    ~ScratchScope() {
      if (outer) release_tune_scratch();
    }
  } scratch_scope{!t_in_slab};
  // one call at a time per matrix: plan fields, the per-matrix timings and the carry buffers of flat / row-block-plus belong
  // to the plan (two host threads on DIFFERENT matrices do not meet here; this lock is never held together with g_mu)
  std::lock_guard<std::mutex> plan_lock(p->mu);
  t_beta_class = beta != 0.0 ? 1 : 0;
  // this call's tuning budget (see defer_tuning): the first call on a matrix may spend first_call_budget SpMV-equivalents on trial launches,
  // a later one later_call_budget while something is still open; spmv_acc_prepare, captures (which time nothing) and `deterministic` are outside it
  struct BudgetScope {
    Plan &p;
    bool outer;
    int strategy, cls;
    ~BudgetScope() {
      if (!outer) return;
      if (t_first_trial_ms > 0.f) p.trial_ms = t_first_trial_ms;
      p.tuning_open = t_tuning_deferred; // (a call that deferred nothing has settled everything on its path)
      if (!t_tuning_deferred && strategy >= 0 && strategy < 32) p.settled_for[cls] |= 1u << strategy;
      t_budget_spmvs = 0.0;
      t_budget_ms = -1.0;
    }
  } budget_scope{*p, !t_in_slab, strategy_named, beta != 0.0 ? 1 : 0};
  if (!t_in_slab) {
    // (the rule-twin path: the FIRST call's budget covers the twin's build and serve too; a later call's two SpMV-equivalents start here, or a small
    // matrix' twin launch alone would use them up and no phase would ever start)
    t_call_began = (t_early_clock && p->calls == 0) ? t_early_began : std::chrono::steady_clock::now();
    t_tuning_deferred = false;
    t_first_trial_ms = 0.f;
    const int allowance = p->calls == 0 ? tun(kT_first_call_budget) : tun(kT_later_call_budget);
    t_budget_spmvs = (t_unbounded_tuning > 0 || t_capturing || allowance <= 0) ? 0.0 : static_cast<double>(allowance);
    t_budget_ms = (t_budget_spmvs > 0.0 && p->trial_ms > 0.0) ? t_budget_spmvs * p->trial_ms : -1.0;
  }
  // What the reference's harness calls `pre` (its per-call break-point / analysis cost, benchmark_time.cpp:23-43) is paid here
  // by the FIRST call on a matrix: structural passes + per-matrix timings, all of which end in a synchronisation, so the host
  // time from here to the return of that call is the preparation time (the final launch itself is asynchronous).
  // A later call that builds another family's plan (first flat call after adaptive-plus calls, a changed tunable) counts too.
  struct PrepareClock {
    unsigned work0;
    std::chrono::steady_clock::time_point t0;
    ~PrepareClock() {
      t_last_prepare_us =
          t_plan_work != work0 ? std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() : 0.0;
    }
  } prepare_clock{t_plan_work, std::chrono::steady_clock::now()};
  if (p->calls++ == 0) ++t_plan_work; // the plan itself (nnz / guard samples) was just made by get_plan
  (void)hipGetLastError(); // errors of earlier, unrelated HIP calls of this thread are not this call's
  // A plan owns scratch that its kernels write (flat's carries, row-block-plus partials, the slab passes' partial sums, LIGHT's counter):
  // two SpMVs of one matrix in flight on DIFFERENT streams would share it.  The lock above orders the enqueueing, this orders the execution:
  // a call on another stream than the plan's last one waits for that one's work (an event behind its last launch).  Same stream: nothing.
  if (p->launched && p->last_stream != st && !t_capturing && !t_in_slab) {
    if (!p->order_event && hipEventCreateWithFlags(&p->order_event, hipEventDisableTiming) != hipSuccess) p->order_event = nullptr;
    if (p->order_event && hipEventRecord(p->order_event, p->last_stream) == hipSuccess) (void)hipStreamWaitEvent(st, p->order_event, 0);
    (void)hipGetLastError(); // (the other stream may have been destroyed by its owner: its work is then complete)
  }
  p->last_stream = st;
  p->launched = true;
  if (t_capturing) p->captured = true;
  // where this call's kernels read the old y (kernels.hpp CsrDev::yin); the plan's lock is held until the launches are enqueued
  struct YinScope {
    CsrDev &A;
    ~YinScope() { A.yin = nullptr; }
  } yin_scope{p->A};
  p->A.yin = beta != 0.0 ? dy_in : nullptr;
  p->last_c16 = 0; // (run_rowblock / run_flat say otherwise when this call's kernel reads the 16-bit column encoding)

  if (p->A.count() == 0) {
    launch_scale_y(st, m, beta, dy, p->A.yin);
    p->last_kernel = kKernelScaleOnly;
    return;
  }
  // tunable strict_strategy: the name the caller gave binds the kernel (run_flat, run_plus); a slab of the opt-in column slabs keeps its parent's
  struct StrictScope {
    int prev;
    ~StrictScope() { t_strict_name = prev; }
  } strict_scope{t_strict_name};
  if (!t_in_slab) t_strict_name = tun(kT_strict_strategy) ? strategy : -1;
  if (!d_colindex || !d_value) {
    set_error(kErrBadArgument, "null colindex / value with nnz > 0");
    return;
  }
  if (tun(kT_validate) && !validate_plan(*p, st)) return;
  if (tun(kT_guard_full) && !t_in_slab && !launch_full_guard(*p, st)) return; // (a slab is a derived matrix: its parent was checked)

  int copy_slabs = (tun(kT_col_slabs) >= 2 && !t_in_slab) ? (tun(kT_col_slabs) > 64 ? 64 : tun(kT_col_slabs)) : 0; // (forced: up to 64 slabs -- one lane of the build's wavefront per slab)
  if (!copy_slabs && tun(kT_col_slabs) < 0 && !t_in_slab) copy_slabs = slab_copy_auto(*p, strategy, st, n, dx); // (round 6: the automatic slab-major copy)
  if (copy_slabs) {
    if (!run_col_slabs(*p, copy_slabs, strategy, st, alpha, beta, m, n, dx, dy)) return;
    t_last_plan = p;
    t_beta_class = beta != 0.0 ? 1 : 0;
    if (t_plan_work != prepare_clock.work0 && p->tune_key) tune_store(*p);
    return;
  }

  if (tun(kT_slab_segments) >= 1 && !t_in_slab && !(t_strict_name == kLineEnhance || t_strict_name == kLine || t_strict_name == kFlat)) {
    // column-slab blocking without a copy: S passes over the plan's run lists (k_segment.hip), whatever the strategy name
    // (1 = the AUTOMATIC slab count -- the x-size rule, seg_auto_slabs -- with the passes always taken: what the timed choice runs where it wins)
    const int S = tun(kT_slab_segments) == 1 ? seg_auto_slabs(n) : tun(kT_slab_segments) > 16 ? 16 : tun(kT_slab_segments); // (ensure_segments takes one off when the whole-row plane of the two-class form needs it: 16 planes in all)
    if (last_error_code_only() == kOk && !t_capturing && !ensure_segments(*p, S, st)) {
      // (no room for the lists or their S x (m + 1) build temporaries: the passes are an optimisation, the strategy's own kernel runs)
      (void)hipGetLastError();
      tune_log("m %d nnz %d: slab_segments: the run lists could not be built (%s), ordinary path", m, p->A.nnz, last_error_string());
      clear_error();
      p->free_segments();
      p->seg_state = 0; // (not tried again for this plan)
    }
    if (p->seg_state == 1) {
      run_segments(st, *p, alpha, beta, dx, dy);
      if (t_plan_work != prepare_clock.work0 && p->tune_key) tune_store(*p);
      return;
    }
    // (rows not ordered by column slab: the ordinary path below)
  }


  const long long avg = static_cast<long long>(p->A.count()) / m;
  // a resident grid for the two persistent-style legacy kernels: CUs x 8 workgroups of 4 waves
  auto resident_blocks = [&]() {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, p->device) != hipSuccess || cus <= 0) cus = 256;
    (void)hipGetLastError();
    return cus * 8;
  };
  if (strategy == kLight) {
    // LightSpMV (hip-light/light_spmv.cpp:16-41): lanes per row from the average row length (its thresholds: vector_row.cpp's
    // table), rows handed out by the plan's counter
    if (!p->d_light_counter) {
      if (!plan_work_allowed("LIGHT's row counter")) return;
      ++t_plan_work;
      if (!hip_ok(hipMalloc(reinterpret_cast<void **>(&p->d_light_counter), 2 * sizeof(unsigned)), "hipMalloc light counter") ||
          !hip_ok(hipMemsetAsync(p->d_light_counter, 0, 2 * sizeof(unsigned), st), "memset light counter") ||
          !hip_ok(hipStreamSynchronize(st), "sync light counter"))
        return;
    }
    launch_light(st, p->A, classic_vec(avg), resident_blocks(), p->d_light_counter, alpha, beta, dx, dy);
    p->last_kernel = kKernelLight;
    strategy = -1; // handled
  } else if (strategy == kBlockRowOrdinary) {
    launch_block_row(st, p->A, resident_blocks(), alpha, beta, dx, dy); // hip-block-row-ordinary/spmv_hip_acc_imp.cpp:16-75
    p->last_kernel = kKernelBlockRow;
    strategy = -1;
  }
  switch (strategy) {
  case -1:
    break;
  case kVectorRow: // (LIGHT and BLOCK_ROW_ORDINARY were served above: k_legacy.hip, what the names mean in the reference)
  {
    const int forced_w = tun(kT_vector_width);
    const bool tile_form = tun(kT_vector_tile) != 0;
    const int w = (forced_w >= 1 && forced_w <= 64 && (forced_w & (forced_w - 1)) == 0) ? forced_w
                  : tile_form ? tile_vec(avg) : classic_vec(avg);
    if (tun(kT_rowblock_guard) && !probe_rowblock(*p, kThreads / w, st)) return;
    if (tile_form && p->rowblock_ok == 0) {
      // very uneven rows (hub rows of a power-law matrix): w lanes walking a row of 10^5 non-zeros serialise the kernel (5.8 ms
      // on a 60 000-row power-law matrix that the other families run in 30 us); same rescue as the row-block family
      run_plus(st, *p, h_rowptr, alpha, beta, dx, dy);
    } else if (tile_form && p->rowblock_ok != 0) {
      // the reference's lane width per row (vector_row.cpp:15-27) on the tile machinery
      const double a = static_cast<double>(p->A.count()) / m;
      auto launch = [&](int pol, double al, double be, double *yy) {
        launch_vector_tile(st, p->A, m, w, w, a, a, kVectorTarget, tun(kT_xcd_chunk), pol, al, be, dx, yy,
                           next_reverse(*p));
      };
      if (!autotune_policy(*p, kFamVector, st, [&](int pol, double *ys) { launch(pol, 1.0, trial_beta(), ys); })) return;
      launch(policy_for(*p, kFamVector), alpha, beta, dy);
      p->last_kernel = kKernelVectorTile;
    } else {
      launch_vector_row(st, p->A, m, w, 1, alpha, beta, dx, dy, p->rowblock_ok == 0);
      p->last_kernel = kKernelVectorRow;
    }
    break;
  }
  case kWfRow:
    launch_wave_row(st, p->A, alpha, beta, dx, dy);
    p->last_kernel = kKernelWaveRow;
    break;
  case kDefault: // the reference's DEFAULT is its one-lane sequential correctness kernel (hip/spmv_hip_acc_imp.cpp:15-35) and
                 // also what its build ships with (config.cmake:15): here the name gets the general-purpose kernel
    run_rowblock(st, *p, h_rowptr, alpha, beta, dx, dy, true); // engine's choice, like adaptive's even-matrix branch
    break;
  case kThreadRow:
    // KERNEL_STRATEGY THREAD_ROW (hip-thread-row/thread_row.cpp:17-48, thread_row_block.hpp): ONE lane sums each row -- the reference's
    // block-level form: the workgroup's non-zeros staged through LDS by coalesced loads, then a thread per row.  Up to 5.8 non-zeros
    // per row that is the row-block kernel's own shape; beyond, where the reference falls back to a 128-block naive loop, the name
    // keeps its meaning here (rows longer than a wavefront are handed to whole waves by tile_row_sum)
    run_rowblock(st, *p, h_rowptr, alpha, beta, dx, dy, false, 1);
    break;
  case kLineEnhance:
  case kLine:
    run_rowblock(st, *p, h_rowptr, alpha, beta, dx, dy);
    break;
  case kFlat:
    run_flat(st, *p, alpha, beta, dx, dy);
    break;
  case kAdaptive: {
    if (!fetch_samples(*p, h_rowptr)) return;
    if (tun(kT_adaptive_timed) && !tun(kT_adaptive_split) && !tun(kT_deterministic)) {
      run_adaptive_timed(st, *p, h_rowptr, alpha, beta, dx, dy);
      break;
    }
    // untimed form: the reference's decision tree on four rowptr samples, re-targeted at this library's kernels
    switch (adaptive_branch(m, p->samples)) {
    case 1:
      // The two row halves differ >= 4x in non-zeros.  The reference answers with two lane widths, one per half
      // (vector_row.cpp:30-38; still available as adaptive_vec_row_sparse_spmv / tunable adaptive_split).  Blocks cut by
      // non-zero count with lanes per row chosen per block fit such a matrix better: on a 2 M-row matrix with halves of 40
      // and 5 nnz/row the split took 185 us, row-block-plus 110 us, flat 110 us, fixed row blocks 120 us.
      if (tun(kT_adaptive_split)) {
        const int half_rows = m / 2;
        const long long a0 = half_rows > 0 ? p->samples.half / half_rows : 0;
        const long long a1 = (static_cast<long long>(p->samples.last) - p->samples.half) / (m - half_rows);
        if (tun(kT_vector_tile)) {
          const double f0 = half_rows > 0 ? static_cast<double>(p->samples.half) / half_rows : 0.0;
          const double f1 = (static_cast<double>(p->samples.last) - p->samples.half) / (m - half_rows);
          launch_vector_tile(st, p->A, half_rows, tile_vec(a0), tile_vec(a1), f0, f1, kVectorTarget,
                             tun(kT_xcd_chunk), policy_for(*p, kFamVector), alpha, beta, dx, dy);
        } else {
          launch_vector_row(st, p->A, half_rows, classic_vec(a0), classic_vec(a1), alpha, beta, dx, dy);
        }
      } else {
        run_plus(st, *p, h_rowptr, alpha, beta, dx, dy);
      }
      break;
    default:
      // 2 (adaptive line), 3 (adaptive line-enhance), 4 (adaptive flat), 5 (line-enhance).  The reference sends
      // branch 4 (nnz > 2^23) to flat because its row-block kernels lose balance on large irregular matrices; here
      // the row-block kernel carries a plan-time balance probe and falls back to the row-block-plus kernel exactly
      // then, and measures 1-5 % faster than flat on the balanced large-set stand-ins (one kernel, no carry
      // fix-up), so every non-split branch goes through it.
      // Fixed row blocks are sized from the matrix-wide average row length; where the four row quarters (the samples the
      // decision already holds) differ 1.75x or more in non-zeros, blocks cut by non-zero count fit better: row-block-plus
      // measures 3-7 % faster at 2x-3x (tools/halves_bench.py), the same within 1 % at 1.5x.
      if (quarters_uneven(p->samples) && !tun(kT_adaptive_split)) run_plus(st, *p, h_rowptr, alpha, beta, dx, dy);
      else run_rowblock(st, *p, h_rowptr, alpha, beta, dx, dy, true);
      break;
    }
    break;
  }
  case kAdaptivePlus:
    run_plus(st, *p, h_rowptr, alpha, beta, dx, dy);
    break;
  default:
    set_error(kErrUnknownStrategy, "unknown strategy id");
    break;
  }
  if (t_plan_work != prepare_clock.work0 && p->tune_key) tune_store(*p); // (plan work happened: keep what was learnt)
  // a launch that failed (bad grid, no code object for this device) leaves y untouched: say so
  const hipError_t launch_err = hipGetLastError();
  if (launch_err != hipSuccess && last_error_code_only() == kOk)
    set_error(kErrHip, std::string("kernel launch failed: ") + hipGetErrorString(launch_err));
}
} // namespace

} // namespace spmv_acc
