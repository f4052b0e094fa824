"""GPU suite (-m gpu): size-selected branches at test size, strict strategy names, chunk views, the tune cache's provisional choice, the driver-facing timings (until round 6: test_gpu_round5.py) --
  * every SIZE-SELECTED branch of the plan and the launchers at test size (the round-4 regression -- 17 planes once x >= 496 MB -- passed 153 tests and
    was found by an out-of-suite R-MAT 26 probe): the size rules are tunables now (slab_kb, hint_min_x_mb, max_grid_blocks, flat_small_nnz_k) and this
    file crosses each of them on matrices of 10^4 .. 10^5 rows, against the CPU oracle (tests/size_thresholds.py is the registry the CPU suite checks);
  * tunable strict_strategy: a strategy name means its algorithm (spmv_acc_query_plan_last_kernel);
  * the tune cache never stores adaptive's PROVISIONAL choice (advisor, round 4);
  * un-rebased row sub-ranges (a pipelined shard's chunk views) are planned by THEIR OWN non-zeros (advisor, round 4);
  * bench.py's driver-facing numbers: ms_per_step is the kernel's time (median of repeated regions).
Same tolerances as tests/test_gpu_parity.py: scaled error <= 1e-12 against the oracle (cli/verification.cpp:56-66), bit-exact where two paths must agree."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import spmv_acc_amd
from spmv_acc_amd import synth

pytestmark = pytest.mark.gpu

SCALED_TOL = 1e-12
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def torch_dev(hiplib):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _sorted_rows(rowptr, cols, vals):
    cols, vals = cols.copy(), vals.copy()
    for i in range(len(rowptr) - 1):
        a, b = rowptr[i], rowptr[i + 1]
        order = np.argsort(cols[a:b], kind="stable")
        cols[a:b] = cols[a:b][order]
        vals[a:b] = vals[a:b][order]
    return cols, vals


def _check(torch, oracle, hiplib, mat, tunables, strategies, tag, alpha=0.5, beta=-2.0, want_kernel=None, calls=3):
    """One matrix under one set of tunables through `strategies`: every call (the first builds and times, the later ones finish what the call
    budget left open) against the oracle."""
    rowptr, cols, vals, n = mat
    m, nnz = len(rowptr) - 1, int(rowptr[-1])
    rng = np.random.default_rng(5)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    try:
        for k, v in tunables.items():
            assert hiplib.spmv_acc_set_tunable(k.encode(), v) == 0, k
        for strat in strategies:
            for call in range(calls):
                y = dy0.clone()
                spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y, strategy=strat)
                torch.cuda.synchronize()
                err = oracle.scaled_error(y.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0)
                assert err <= SCALED_TOL, (tag, tunables, strat, call, err)
            info = spmv_acc_amd.query_plan(drp, m)
            if want_kernel is not None:
                assert info["last_kernel"] in want_kernel, (tag, tunables, strat, info)
            spmv_acc_amd.release_plans(drp)
        return info
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def _powerlaw_sorted(m, n, seed):
    """Power-law columns, column-sorted rows, many rows of EXACTLY 32 non-zeros (the rows the one-lane count kernel cuts by slab: the round-4 bug)."""
    rng = np.random.default_rng(seed)
    lens = rng.choice([32, 32, 32, 31, 33, 7, 3, 64, 200, 0], size=m)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    cols = np.minimum((rng.random(nnz) ** 5 * n).astype(np.int64), n - 1).astype(np.int32)
    vals = rng.standard_normal(nnz)
    cols, vals = _sorted_rows(rowptr, cols, vals)
    return rowptr, cols, vals, n


# ---- every size-selected branch, at test size ------------------------------------------------------------------------------------------------
def test_automatic_slab_count_at_its_maximum(torch_dev, oracle, hiplib):
    """The x-size rule of the column-slab passes (seg_auto_slabs: one slab per `slab_kb` of x, 2 .. 16; 16 from x = 496 MB on) pushed to its maximum
    on a 90,000-column matrix, with the passes always taken (slab_segments = 1: the automatic COUNT, no timing), two-class form on and off:
    16 planes in all, never 17 (tuner.cpp ensure_segments, kernels.hpp kSegMaxPlanes).  With the round-4 clamp removed this test fails."""
    mat = _powerlaw_sorted(40000, 90000, 77)
    for whole_below, want in ((32, 15), (0, 16), (64, 15)):
        info = _check(torch_dev, oracle, hiplib, mat, {"slab_segments": 1, "slab_kb": 32, "slab_whole_below": whole_below},
                      ("line_enhance", "flat", "adaptive"), "auto slab count", want_kernel=("slab_passes",))
        assert info["slab_passes"] == want, (whole_below, info)
    # fewer slabs from the same rule
    info = _check(torch_dev, oracle, hiplib, mat, {"slab_segments": 1, "slab_kb": 128}, ("line_enhance",), "auto slab count 5", want_kernel=("slab_passes",))
    assert info["slab_passes"] == 5, info  # (720,000 B of x + half a slab) // 128 KB


def test_automatic_slab_passes_through_the_timed_choice(torch_dev, oracle, hiplib):
    """The automatic mode as shipped (slab_segments = -1) with its size rules lowered to test size: the column census runs (hint_min_x_mb 0), finds a hot
    set on power-law columns (a budget of 64 KB of x lines), the run lists are built with the maximum slab count and timed against the row-block-plus
    kernel; whichever wins runs -- every call, also those that finish the timings under the call budget, matches the oracle."""
    mat = _powerlaw_sorted(60000, 200000, 3)
    for extra in ({}, {"first_call_budget": 1}, {"slab_whole_below": 0}):
        tun = dict({"slab_kb": 64, "hint_min_x_mb": 0, "hint_budget_kb": 64}, **extra)
        _check(torch_dev, oracle, hiplib, mat, tun, ("line_enhance", "adaptive_plus", "adaptive", "flat"), "auto timed", calls=4,
               want_kernel=("slab_passes", "rowblock_plus", "rowblock", "flat_tile"))


def test_automatic_slab_major_copy_and_its_value_guard(torch_dev, oracle, hiplib):
    """Round 6: a plan whose own timed choice is the slab passes builds, inside spmv_acc_prepare or after 32 calls, the slab-major
    COPY (k_slab.hip), times it against the passes and keeps the faster (tunable col_slabs = -1, the default; -2 keeps the copy whatever the timing says,
    which is how this test pins the path at test size).  The copy holds VALUES: before every use 65,536 samples of the caller's values are compared with
    the copy's and the copy is refreshed when they differ -- a caller who rewrites the values in place is served the NEW values by the very next call,
    without an error.  `deterministic`, `strict_strategy`, col_slabs = 0 and stream captures keep to the passes (which hold no values at all)."""
    torch = torch_dev
    rowptr, cols, vals, n = _powerlaw_sorted(60000, 200000, 3)
    m, nnz = len(rowptr) - 1, int(rowptr[-1])
    rng = np.random.default_rng(7)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    # (slab_segments = 1: the run-list passes with the automatic slab count whatever the timing against row-block-plus says -- at test size the passes do
    # not win by themselves; with col_slabs = -2 the copy then replaces them as it replaces a plan's own timed choice)
    size_rules = {"slab_segments": 1, "slab_kb": 64}

    def spmv(values, alpha=0.5, beta=-2.0, expect=None):
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y, strategy="line_enhance")
        torch.cuda.synchronize()
        ref = oracle.host_spmv(alpha, beta, rowptr, cols, values, x, y0)
        assert oracle.scaled_error(y.cpu().numpy(), ref, alpha, beta, rowptr, cols, values, x, y0) <= SCALED_TOL
        assert hiplib.spmv_acc_last_error() == 0
        kernel = spmv_acc_amd.query_plan(drp, m)["last_kernel"]
        if expect is not None:
            assert kernel in expect, (kernel, expect)
        return kernel

    try:
        for k, v in dict(size_rules, col_slabs=-2).items():
            assert hiplib.spmv_acc_set_tunable(k.encode(), v) == 0
        spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy="line_enhance")
        assert spmv_acc_amd.query_plan(drp, m)["slab_passes"] >= 2
        spmv(vals, expect=("col_slabs",))
        # every value rewritten in place: the next call sees it, refreshes the copy and computes with the new values
        vals2 = vals * -0.75 + 0.125
        dv.copy_(dev(torch, vals2))
        spmv(vals2, expect=("col_slabs",))
        spmv(vals2, alpha=1.0, beta=0.0, expect=("col_slabs",))
        # a contiguous tenth of the values (wider than the sample spacing of nnz / 65,536): noticed as well
        vals3 = vals2.copy()
        vals3[nnz // 3: nnz // 3 + nnz // 10] *= 3.0
        dv.copy_(dev(torch, vals3))
        spmv(vals3, expect=("col_slabs",))
        # column indices rewritten in place (same rowptr): the copy's structure is stale -- it is dropped and the run-list passes, which read the caller's
        # arrays, serve this call and the following ones; then everything back for the rest of the test
        cols_b = np.concatenate([np.sort(n - 1 - cols[rowptr[i]:rowptr[i + 1]]) for i in range(m)]).astype(np.int32)
        dci.copy_(dev(torch, cols_b))
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, y, strategy="line_enhance")
        torch.cuda.synchronize()
        ref_b = oracle.host_spmv(0.5, -2.0, rowptr, cols_b, vals3, x, y0)
        assert oracle.scaled_error(y.cpu().numpy(), ref_b, 0.5, -2.0, rowptr, cols_b, vals3, x, y0) <= SCALED_TOL
        assert spmv_acc_amd.query_plan(drp, m)["last_kernel"] == "slab_passes"
        dci.copy_(dev(torch, cols))
        spmv_acc_amd.release_plans(drp)
        spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy="line_enhance")
        spmv(vals3, expect=("col_slabs",))
        # inside a capture the passes serve (the guard needs a synchronisation): same result up to the order of the sums
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            ys = dy0.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, ys, strategy="line_enhance")
            ys.copy_(dy0)
            g.replay()
        torch.cuda.synchronize()
        ref = oracle.host_spmv(0.5, -2.0, rowptr, cols, vals3, x, y0)
        assert oracle.scaled_error(ys.cpu().numpy(), ref, 0.5, -2.0, rowptr, cols, vals3, x, y0) <= SCALED_TOL
        spmv_acc_amd.release_plans(drp)
        # the switches that keep the copy out
        for off in ({"col_slabs": 0}, {"deterministic": 1}, {"strict_strategy": 1}):
            hiplib.spmv_acc_reset_tunables()
            for k, v in {**size_rules, "col_slabs": -2, **off}.items():
                assert hiplib.spmv_acc_set_tunable(k.encode(), v) == 0
            spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy="line_enhance")
            assert spmv(vals3) != "col_slabs", off
            spmv_acc_amd.release_plans(drp)
        # outside spmv_acc_prepare the copy is built by the call after the 32nd: passes until then, the copy from then on
        hiplib.spmv_acc_reset_tunables()
        for k, v in dict(size_rules, col_slabs=-2).items():
            assert hiplib.spmv_acc_set_tunable(k.encode(), v) == 0
        kernels = [spmv(vals3, expect=("col_slabs", "slab_passes")) for _ in range(36)]
        assert kernels[:32] == ["slab_passes"] * 32 and kernels[-1] == "col_slabs", kernels
        # the default (-1) decides by timing, and only for a plan whose OWN timed choice is the passes: whatever runs matches the oracle
        spmv_acc_amd.release_plans(drp)
        hiplib.spmv_acc_reset_tunables()
        for k, v in {"slab_kb": 64, "hint_min_x_mb": 0, "hint_budget_kb": 64}.items():
            assert hiplib.spmv_acc_set_tunable(k.encode(), v) == 0
        spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy="line_enhance")
        for _ in range(3):
            spmv(vals3, expect=("col_slabs", "slab_passes", "rowblock_plus", "rowblock"))
    finally:
        hiplib.spmv_acc_set_stream(None)
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def test_grid_stride_paths_at_test_size(torch_dev, oracle, hiplib):
    """Kernels whose grid grows with m stride over the rows beyond kMaxGridBlocks workgroups (8,388,593: 33.5 M rows at one wavefront per row; round 3
    found two of them returning wrong results at 70 M rows).  With the cap lowered to 256 workgroups the striding runs at 10^5 rows: wf_row /
    block_row_ordinary's stand-in (wave_row_kernel), the direct vector-row form, the column-slab count / scatter kernels, the run-list count kernel."""
    rowptr, cols, vals = synth.random_csr(100_000, 100_000, 6, seed=9, kind="powerlaw")
    cols, vals = _sorted_rows(rowptr, cols, vals)
    mat = (rowptr, cols, vals, 100_000)
    cap = {"max_grid_blocks": 256}
    _check(torch_dev, oracle, hiplib, mat, cap, ("wf_row",), "wave rows", want_kernel=("wave_row",))
    _check(torch_dev, oracle, hiplib, mat, dict(cap, vector_tile=0, vector_width=64), ("vector_row",), "direct vector rows", want_kernel=("vector_row",))
    _check(torch_dev, oracle, hiplib, mat, dict(cap, vector_tile=0, vector_width=2, rowblock_guard=0), ("vector_row",), "narrow vector rows",
           want_kernel=("vector_row",))
    _check(torch_dev, oracle, hiplib, mat, dict(cap, col_slabs=4), ("line_enhance",), "column slabs", want_kernel=("col_slabs",))
    _check(torch_dev, oracle, hiplib, mat, dict(cap, slab_segments=5), ("line_enhance",), "run lists", want_kernel=("slab_passes",))


def test_flat_size_rules_and_hypersparse_tiles(torch_dev, oracle, hiplib):
    """flat's size rules: (a) below flat_small_nnz_k Ki non-zeros (24 Mi: small grids) tile size and staging order are timed per matrix, above they are
    not -- both sides of the rule on one 1.2 M-non-zero matrix; (b) a tile that owns more than 16,384 rows (hypersparse: 300,000 rows, 2,000
    non-zeros -- all in ONE tile) hands the matrix to the row-block kernel, and under strict_strategy runs the tile kernel itself."""
    rowptr, cols, vals = synth.random_csr(150_000, 150_000, 8, seed=4, kind="uniform")
    mat = (rowptr, cols, vals, 150_000)
    for k in (1, 24 << 10):
        _check(torch_dev, oracle, hiplib, mat, {"flat_small_nnz_k": k, "flat_rowblock": 0}, ("flat",), "flat small-grid rule", want_kernel=("flat_tile",))
    rng = np.random.default_rng(8)
    m = 300_000
    lens = np.zeros(m, dtype=np.int64)
    lens[rng.choice(m, 1000, replace=False)] = 2
    hrp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    hci = rng.integers(0, m, int(hrp[-1])).astype(np.int32)
    hv = rng.standard_normal(int(hrp[-1]))
    hyper = (hrp, hci, hv, m)
    _check(torch_dev, oracle, hiplib, hyper, {}, ("flat",), "hypersparse", want_kernel=("rowblock", "rowblock_plus"))
    _check(torch_dev, oracle, hiplib, hyper, {"strict_strategy": 1}, ("flat",), "hypersparse strict", want_kernel=("flat_tile",), calls=2)


def test_row_digest_rule_and_its_long_row_escape(torch_dev, oracle, hiplib):
    """The row digest (1-byte row lengths instead of rowptr) is taken where rows average <= 8 non-zeros; a block holding a row longer than 255 reads
    rowptr after all (bit 31 of its base).  Both sides of the average rule, and the escape, on 80,000-row matrices."""
    rng = np.random.default_rng(2)
    for avg, spike in ((5, 0), (5, 300), (12, 0), (12, 300)):
        lens = rng.poisson(avg, 80_000)
        if spike:
            lens[rng.choice(80_000, 40, replace=False)] = rng.integers(256, 2 * spike, 40)
        rowptr, cols, vals = synth.csr_from_row_lengths(lens, 80_000, rng, locality=64, far_fraction=0.05)
        _check(torch_dev, oracle, hiplib, (rowptr, cols, vals, 80_000), {}, ("line_enhance", "adaptive", "thread_row"), f"row digest avg {avg} spike {spike}",
               want_kernel=("rowblock", "rowblock_plus", "flat_tile"))


def test_x_beyond_the_hinted_gathers_reach(torch_dev, oracle, hiplib):
    """Hinted gathers address x by 32-bit byte offsets: with 8 * n >= 4 GB the plan must not build or use hints even when they are forced
    (gather_hint = 1).  A 20,000-row matrix whose columns spread over an x of 2^29 + 2^27 entries (5.4 GB); the oracle works on the columns the
    matrix references (renumbered), which is the same arithmetic."""
    torch = torch_dev
    m, n = 20000, (1 << 29) + (1 << 27)  # 5.4 GB of x: a fifth of the columns lie beyond byte offset 2^32
    rng = np.random.default_rng(4)
    lens = rng.integers(3, 12, m)
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    cols64 = rng.integers(0, n, nnz)
    cols64 = np.concatenate([np.sort(cols64[rowptr[i]:rowptr[i + 1]]) for i in range(m)])
    vals = rng.standard_normal(nnz)
    used, compact = np.unique(cols64, return_inverse=True)
    xs = rng.standard_normal(len(used))
    y0 = rng.standard_normal(m)
    ref = oracle.host_spmv(0.5, -2.0, rowptr, compact.astype(np.int32), vals, xs, y0)
    dx = torch.zeros(n, dtype=torch.float64, device="cuda")
    dx[torch.from_numpy(used).cuda()] = torch.from_numpy(xs).cuda()
    drp, dci, dv, dy0 = (dev(torch, a) for a in (rowptr, cols64.astype(np.int32), vals, y0))
    assert int((cols64.astype(np.int64) * 8 >= (1 << 32)).sum()) > nnz // 8
    try:
        for tun in ({}, {"gather_hint": 1}, {"gather_hint": 1, "slab_segments": 3}):
            for k, v in tun.items():
                hiplib.spmv_acc_set_tunable(k.encode(), v)
            for strat in ("line_enhance", "adaptive_plus", "flat", "adaptive"):
                y = dy0.clone()
                spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, y, strategy=strat)
                torch.cuda.synchronize()
                err = oracle.scaled_error(y.cpu().numpy(), ref, 0.5, -2.0, rowptr, compact.astype(np.int32), vals, xs, y0)
                assert err <= SCALED_TOL, (tun, strat, err)
            hiplib.spmv_acc_reset_tunables()
            spmv_acc_amd.release_plans(drp)
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()
        del dx
        torch.cuda.empty_cache()


# ---- strict_strategy -----------------------------------------------------------------------------------------------------------------------------
def test_strict_strategy_runs_the_named_kernel(torch_dev, oracle, hiplib):
    """strategy_picker.cpp:19-65: in the reference the strategy name IS the kernel.  Here a name selects a policy by default (flat may run the row-block
    kernel where it timed faster; line_enhance the column-slab passes on power-law columns) and `strict_strategy = 1` binds the name to its algorithm:
    flat -> flat_tile_kernel, line_enhance / line -> the row-block kernel (row-block-plus where fixed row blocks are unbalanced), whatever was timed
    or forced.  spmv_acc_query_plan_last_kernel tells which kernel ran."""
    torch = torch_dev
    # (a) balanced FEM-like rows: flat's substitution forced ON, then overridden by strict
    rowptr, cols, vals = synth.random_csr(120_000, 120_000, 20, seed=6, kind="uniform")
    fem = (rowptr, cols, vals, 120_000)
    _check(torch, oracle, hiplib, fem, {"flat_rowblock": 1}, ("flat",), "flat, policy", want_kernel=("rowblock",))
    _check(torch, oracle, hiplib, fem, {"flat_rowblock": 1, "strict_strategy": 1}, ("flat",), "flat, strict", want_kernel=("flat_tile",))
    _check(torch, oracle, hiplib, fem, {"strict_strategy": 1}, ("flat",), "flat, strict, default tunables", want_kernel=("flat_tile",))
    _check(torch, oracle, hiplib, fem, {"strict_strategy": 1}, ("line_enhance", "line"), "row blocks, strict", want_kernel=("rowblock",))
    # (b) power-law columns with the slab passes forced: the named strategies keep their own kernels under strict, adaptive (the engine's choice) does not care
    pl = _powerlaw_sorted(40000, 90000, 5)
    _check(torch, oracle, hiplib, pl, {"slab_segments": 4}, ("line_enhance", "flat"), "forced passes", want_kernel=("slab_passes",))
    _check(torch, oracle, hiplib, pl, {"slab_segments": 4, "strict_strategy": 1}, ("line_enhance", "line"), "forced passes, strict",
           want_kernel=("rowblock", "rowblock_plus"))
    _check(torch, oracle, hiplib, pl, {"slab_segments": 4, "strict_strategy": 1}, ("flat",), "forced passes, strict flat", want_kernel=("flat_tile",))
    _check(torch, oracle, hiplib, pl, {"slab_segments": 4, "strict_strategy": 1}, ("adaptive",), "forced passes, adaptive", want_kernel=("slab_passes",))
    # (c) the automatic passes (size rules lowered): never for a strictly named line_enhance
    _check(torch, oracle, hiplib, pl, {"slab_kb": 64, "hint_min_x_mb": 0, "hint_budget_kb": 64, "strict_strategy": 1}, ("line_enhance",),
           "automatic passes, strict", want_kernel=("rowblock", "rowblock_plus"), calls=4)


# ---- un-rebased row sub-ranges are planned by their own non-zeros -------------------------------------------------------------------------------------
def test_chunk_views_are_sized_by_their_own_non_zeros(torch_dev, oracle, hiplib):
    """A pipelined shard's chunk k is the view (rowptr + a, nnz = rowptr[b]) of the shard's arrays (shard.cpp).  Until round 5 every shape heuristic read
    that END offset as the chunk's non-zero count: chunk k of C looked (k + 1) times as dense as it is, flat launched the tiles of all preceding chunks
    (empty), the census covered other chunks' columns.  Now the plan reads rowptr[0] once: the last of four chunks launches a quarter of the tiles, picks
    the same lanes per row as the first, and every strategy matches the oracle on it."""
    torch = torch_dev
    m, n = 200_000, 200_000
    rowptr, cols, vals = synth.random_csr(m, n, 10, seed=12, kind="uniform")
    rng = np.random.default_rng(1)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
    C = 4
    cuts = [m * k // C for k in range(C + 1)]
    tiles, vecs = [], []
    try:
        for k in range(C):
            a, b = cuts[k], cuts[k + 1]
            view = drp[a:]  # un-rebased: rowptr[a] > 0 for k > 0
            sub_rp = (rowptr[a:b + 1] - rowptr[a]).astype(np.int32)
            sub_ci, sub_v = cols[rowptr[a]:rowptr[b]], vals[rowptr[a]:rowptr[b]]
            ref = oracle.host_spmv(0.5, -2.0, sub_rp, sub_ci, sub_v, x, y0[a:b])
            for strat, tun in (("flat", {"flat_rowblock": 0}), ("adaptive", {}), ("line_enhance", {}), ("adaptive_plus", {}), ("vector_row", {})):
                for kk, vv in tun.items():
                    hiplib.spmv_acc_set_tunable(kk.encode(), vv)
                y = dev(torch, y0[a:b].copy())
                for _ in range(2):
                    y.copy_(dev(torch, y0[a:b].copy()))
                    spmv_acc_amd.csr_spmv(0.5, -2.0, b - a, n, int(rowptr[b]), view, dci, dv, dx, y, strategy=strat)
                torch.cuda.synchronize()
                err = oracle.scaled_error(y.cpu().numpy(), ref, 0.5, -2.0, sub_rp, sub_ci, sub_v, x, y0[a:b])
                assert err <= SCALED_TOL, (k, strat, err)
                info = spmv_acc_amd.query_plan(view, b - a)
                if strat == "flat":
                    tiles.append(info["flat_tiles"])
                    vecs.append(info["vec"])
                hiplib.spmv_acc_reset_tunables()
            spmv_acc_amd.release_plans(view)
        own = [(int(rowptr[cuts[k + 1]]) - int(rowptr[cuts[k]])) / 2048.0 for k in range(C)]
        for k in range(C):
            assert own[k] - 1 <= tiles[k] <= own[k] + 2, (k, tiles, own)  # its own tiles (+ the two part-owned at its ends), not those of the chunks before
        assert len(set(vecs)) == 1, vecs  # the same rows per workgroup / lanes per row for every chunk of an evenly filled matrix
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def test_destroying_a_pipelined_shard_keeps_the_callers_whole_shard_plan(torch_dev, hiplib):
    """Chunk 0 of a pipelined shard is a view on the caller's own rowptr pointer: dropping the chunk plans must not drop the plan the caller holds on
    that pointer for the whole shard (advisor, round 4: release by (pointer, rows))."""
    torch = torch_dev
    m = n = 50_000
    rowptr, cols, vals = synth.random_csr(m, n, 6, seed=3, kind="uniform")
    drp, dci, dv = (dev(torch, a) for a in (rowptr, cols, vals))
    dx = torch.ones(n, dtype=torch.float64, device="cuda")
    y = torch.zeros(m, dtype=torch.float64, device="cuda")
    nnz = int(rowptr[-1])
    try:
        spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")  # the caller's whole-matrix plan
        half = m // 2
        yh = torch.zeros(half, dtype=torch.float64, device="cuda")
        spmv_acc_amd.csr_spmv(1.0, 0.0, half, n, int(rowptr[half]), drp, dci, dv, dx, yh, strategy="adaptive")  # a chunk-0-like view on the same pointer
        torch.cuda.synchronize()
        assert spmv_acc_amd.query_plan(drp, m) is not None and spmv_acc_amd.query_plan(drp, half) is not None
        lib = spmv_acc_amd.load_library()
        # what spmv_acc_shard_destroy does for its chunks: release by (pointer, rows) -- reached here through the shard API when RCCL is present
        import ctypes

        comm = ctypes.c_void_p()
        if lib.spmv_acc_rccl_comm_init_all(ctypes.byref(comm), 1, None) == 0:
            shard = ctypes.c_void_p()
            rc = lib.spmv_acc_shard_create(ctypes.byref(shard), comm, spmv_acc_amd.strategy_id("adaptive"), m, m, n, nnz, ctypes.c_void_p(drp.data_ptr()),
                                           ctypes.c_void_p(dci.data_ptr()), ctypes.c_void_p(dv.data_ptr()), 2)
            assert rc == 0, lib.spmv_acc_last_error_string()
            assert lib.spmv_acc_shard_prepare(shard, ctypes.c_double(0.0), ctypes.c_void_p(dx.data_ptr())) == 0
            assert spmv_acc_amd.query_plan(drp, half) is not None
            lib.spmv_acc_shard_destroy(shard)
            lib.spmv_acc_rccl_comm_destroy(comm)
            assert spmv_acc_amd.query_plan(drp, m) is not None, "the caller's whole-shard plan went with the chunk plans"
            assert spmv_acc_amd.query_plan(drp, half) is None
        else:
            lib.spmv_acc_clear_error()
            pytest.skip("no RCCL in the process")
    finally:
        spmv_acc_amd.release_plans()


# ---- the tune cache and adaptive's provisional choice ------------------------------------------------------------------------------------------------
_CHILD = r"""
import sys, json, numpy as np, torch
sys.path.insert(0, {root!r})
import spmv_acc_amd
from spmv_acc_amd import synth
rp, ci, v = synth.structured_csr_torch(600_000, 600_000, 4_200_000, 0xC7, device="cuda")
m = n = 600_000
nnz = int(rp[-1].item())
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
out = {{}}
for call in range({calls}):
    y = torch.ones(m, dtype=torch.float64, device="cuda")
    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy="adaptive")
    torch.cuda.synchronize()
if {prepare}:
    spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="adaptive")
info = spmv_acc_amd.query_plan(rp, m)
print("RESULT " + json.dumps({{"settled": info["settled"], "family": info["adaptive_family"]}}))
"""


def _child(tmp_path, cache, calls, prepare, extra_env=None):
    env = dict(os.environ, SPMV_ACC_TUNE_LOG="1", SPMV_ACC_TUNE_CACHE=cache, **(extra_env or {}))
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT, calls=calls, prepare=prepare)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")]
    return json.loads(res[-1][7:]), r.stderr


def test_tune_cache_never_keeps_a_provisional_adaptive_choice(torch_dev, tmp_path):
    """A short process (one call, a tuning budget of one SpMV-equivalent) leaves adaptive's comparison unfinished: the family that serves it is the
    best of those timed so far.  That provisional family must not be read as final by the next process (round 4 stored it without its flag): the cache
    line carries -1 for it, and the second process finishes the comparison -- its log shows the family timings -- before its plan reports settled."""
    cache = str(tmp_path / "tune.txt")
    first, log1 = _child(tmp_path, cache, calls=1, prepare=False, extra_env={"SPMV_ACC_TUNABLES": "first_call_budget=1,later_call_budget=1"})
    assert not first["settled"], (first, log1[-1500:])  # (the budget left the comparison open: what this test is about)
    lines = [ln.split() for ln in open(cache).read().splitlines() if ln.startswith("spmvacc6 ")]
    assert lines, "the first process stored nothing"
    # fields: tag, key, 8 stream policies, adaptive_family[0], adaptive_family[1], ...
    assert lines[-1][10] == "-1" and lines[-1][11] == "-1", lines[-1]
    second, log2 = _child(tmp_path, cache, calls=1, prepare=True)
    assert "adopted from the tune cache" in log2
    assert "-> family" in log2, log2[-2000:]  # the comparison was (re)done here, not adopted half-done
    assert second["settled"] and second["family"] in (0, 1, 2)
    lines = [ln.split() for ln in open(cache).read().splitlines() if ln.startswith("spmvacc6 ")]
    assert lines[-1][11] == str(second["family"]), (lines[-1], second)  # ... and the FINISHED choice is what the cache holds now
    third, log3 = _child(tmp_path, cache, calls=2, prepare=False)
    assert "-> family" not in log3 and third["family"] == second["family"], log3[-1500:]


# ---- the driver-facing numbers ---------------------------------------------------------------------------------------------------------------------------
def test_ms_per_step_is_the_kernels_time(torch_dev):
    """BENCH_r04: one 20-launch wall-clock region carried 0.43 ms of host time (the timing helper settled the plan -- an allocation and a free -- inside
    it) and the driver-timed value read 14 % below the kernel's.  `ms_per_step` is now the MEDIAN of REGION_REPS repetitions of a region that holds
    the K launches and nothing else (spmv_acc_time_spmv_region): at the driver's own arguments it stays within 3 % of the event time of the same
    launches, and the line says how many repetitions it is the median of."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-legs", "--no-cpu-baseline",
                        "--no-sensitivity"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    b2b = line["roofline"]["back_to_back"]
    full = json.load(open(os.path.join(ROOT, "bench_full.json")))
    ev = full["roofline"]["back_to_back"]["launch_ms_mean"]
    assert line["region_reps"] >= 5 and line["steps"] == 20
    assert line["ms_per_step"] <= 1.03 * ev, (line["ms_per_step"], ev, full.get("ms_per_step_wall_all"), full.get("ms_per_step_events_all"))
    assert abs(line["ms_per_step_events"] - ev) < 1e-9 and b2b["frac"] > 0.3
    assert abs(line["value"] - 2.0 * line["config"]["nnz_per_gpu"] / (line["ms_per_step"] * 1e-3) / 1e9) / line["value"] < 1e-3


def test_kernel_clock_reads_the_kernels_own_time(torch_dev, oracle, hiplib):
    """spmv_acc_time_spmv_kernels: every launch of a call carries its own start / stop events (hipExtLaunchKernelGGL) and the call's kernel time is
    their sum -- below the event pair around the call (which also holds the protocol's floor), above most of it for a kernel of tens of microseconds;
    one launch per SpMV for the row-block kernel, two for flat with carries (tile kernel + fix-up), several for the slab passes; and the clock
    changes no result (the same kernels, launched through the extended entry)."""
    torch = torch_dev
    m = n = 400_000
    rowptr, cols, vals = synth.random_csr(m, n, 12, seed=2, kind="uniform")
    cols, vals = _sorted_rows(rowptr, cols, vals)  # (ascending columns inside every row: what the slab passes' run lists need)
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
    try:
        for strat, tun, want_launches in (("line_enhance", {}, (1,)), ("flat", {"flat_rowblock": 0, "flat_finish": 0}, (2,)),
                                          ("flat", {"flat_rowblock": 0, "flat_finish": 1}, (1,)), ("adaptive", {"slab_segments": 3, "slab_whole_below": 0}, range(4, 8)),  # the guard check + three slab passes (+ merges)
                                          ("adaptive", {"slab_segments": 3}, (2,))):  # rows of ~12 non-zeros: all in the whole-row pass (+ the guard check)
            for k, v in tun.items():
                hiplib.spmv_acc_set_tunable(k.encode(), v)
            y = dy0.clone()
            ev, kn, ln = spmv_acc_amd.time_spmv_kernels(strat, 12, 1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, y0=dy0)
            torch.cuda.synchronize()
            assert oracle.scaled_error(y.cpu().numpy(), ref, 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, strat
            assert all(l in want_launches for l in ln), (strat, tun, ln)
            assert all(0.0 < k_ms <= e_ms for k_ms, e_ms in zip(kn, ev)), (strat, kn, ev)
            assert np.median(kn) >= 0.5 * np.median(ev), (strat, np.median(kn), np.median(ev))  # (a ~35 us kernel under a ~5 us floor)
            plain = dy0.clone()
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, plain, strategy=strat)
            torch.cuda.synchronize()
            assert torch.equal(plain, y), strat
            hiplib.spmv_acc_reset_tunables()
            spmv_acc_amd.release_plans(drp)
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def test_subnormal_products_and_sums_are_not_flushed(torch_dev, oracle, hiplib):
    """fp64 subnormals: the reference's host arithmetic (cli/verification.cpp:56-66) keeps them and so must every kernel family -- values and x of
    ~1e-160 make every product subnormal (~1e-320) and every row sum too.  Sums of subnormals are exact in any order; a product is rounded on its
    own (mul, then add) or inside an fma, at most one unit of the last subnormal place (2^-1074) apart per non-zero.  A flush to zero anywhere
    (loads, products, cross-lane sums, the alpha / beta step, stores) would return exact zeros, thousands of units away."""
    torch = torch_dev
    ulp = float(np.nextafter(0.0, 1.0))
    for kind, m, avg in (("uniform", 20000, 6), ("dense_rows", 1500, 400), ("spikes", 20000, 4)):
        n = m
        rowptr, cols, vals = synth.random_csr(m, n, avg, seed=11, kind=kind)
        nnz = int(rowptr[-1])
        rng = np.random.default_rng(12)
        vals = rng.uniform(0.5, 2.0, nnz) * 1e-160 * rng.choice([-1.0, 1.0], nnz)
        x = rng.uniform(0.5, 2.0, n) * 1e-160
        y0 = rng.uniform(0.5, 2.0, m) * 1e-320
        ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
        assert np.count_nonzero(ref) >= 0.99 * m and np.abs(ref).max() < 2.2250738585072014e-308  # (the case is what it says: all subnormal)
        lens = np.diff(rowptr).astype(np.float64)
        drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
        try:
            for strat in spmv_acc_amd.STRATEGIES:
                for tun in ({}, {"flat_rowblock": 0}) if strat == "flat" else ({},):
                    for k, v in tun.items():
                        hiplib.spmv_acc_set_tunable(k.encode(), v)
                    y = dy0.clone()
                    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy=strat)
                    torch.cuda.synchronize()
                    got = y.cpu().numpy()
                    worst = np.max(np.abs(got - ref) / ((lens + 1.0) * ulp))
                    assert worst <= 1.0, (kind, strat, tun, worst, int(np.count_nonzero(got)))
                    hiplib.spmv_acc_reset_tunables()
                    spmv_acc_amd.release_plans(drp)
        finally:
            hiplib.spmv_acc_reset_tunables()
            spmv_acc_amd.release_plans()
