"""GPU suite (-m gpu): the 16-bit column encoding (round 6; spmv_acc_amd/csrc/k_col16.hip builds it, tile_stage.hpp::stage_products_c16 reads it,
tunable col16: -1 timed per matrix and kernel family, 0 never, 1 always where it can be built, 16 / 32 / 64 also pin the record size).

The reference streams one 4-byte column per non-zero (hip-flat/flat_imp_one_pass.hpp:35-39, hip-line-enhance/line_enhance_spmv_imp.inl:55-62); the
encoding must be invisible: the same products into the same per-row sums.  Checked here against the oracle (scaled error <= 1e-12), against the
same kernel reading colindex, on every record size with and without overflow, on row shards that are not rebased, across the size rules that switch
the encoding off, and for the one hazard it adds -- a plan that holds structure derived from colindex."""
import numpy as np
import pytest

import spmv_acc_amd
from spmv_acc_amd import synth

pytestmark = pytest.mark.gpu

SCALED_TOL = 1e-12


@pytest.fixture(scope="module")
def torch_dev(hiplib):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _cases(rng):
    out = []
    out.append(("fem-like, 2 % far", synth.csr_from_row_lengths(rng.integers(20, 40, size=20000), 400000, rng, locality=300, far_fraction=0.02)))
    out.append(("short rows (row digest), 10 % far", synth.csr_from_row_lengths(rng.integers(3, 8, size=60000), 500000, rng, locality=64, far_fraction=0.10)))
    out.append(("long rows", synth.csr_from_row_lengths(rng.integers(900, 5000, size=300), 200000, rng, locality=90000, far_fraction=0.0)))
    lens = rng.integers(0, 30, size=30000)
    lens[rng.integers(0, 30000, 2000)] = 0  # empty rows, also at block starts
    lens[-40:] = 0                          # ... and an empty tail: blocks whose first group starts at nnz
    out.append(("empty rows and an empty tail", synth.csr_from_row_lengths(lens, 30000, rng, locality=500, far_fraction=0.01)))
    for r in range(4):
        lens = rng.integers(4, 12, size=9000)
        lens[-1] += (r - int(lens.sum())) % 4  # nnz mod 4 == r: the block with the ragged end of the arrays reads colindex
        out.append((f"nnz mod 4 = {r}", synth.csr_from_row_lengths(lens, 9000, rng, locality=100, far_fraction=0.03)))
    return out


@pytest.mark.parametrize("strategy,pins", [("line_enhance", {}), ("line_enhance", {"rowlen": 0}), ("default", {}), ("flat", {"flat_rowblock": 0, "flat_npt": 8, "flat_finish": 1, "flat_early": 0}),
                                           ("flat", {"flat_rowblock": 0, "flat_npt": 8, "flat_finish": 0, "flat_early": 0})])  # (flat's own timed choices pinned: the two plans must sum alike)
@pytest.mark.parametrize("rec", [1, 16, 32, 64])
def test_encoding_matches_colindex(torch_dev, oracle, hiplib, strategy, pins, rec):
    """Forced encoding (every record size: 16 overflows on the 10 % matrix, 64 never does) against the same kernel on colindex and against the oracle;
    flat's tile origins do not depend on the encoding, so there the two are bit-identical; the row blocks' origin moves from a multiple of 4 to a
    multiple of 256, which may split a block's rows over two rounds: equal to a few ulps of the row's terms."""
    torch = torch_dev
    rng = np.random.default_rng(61)
    try:
        for tag, (rowptr, cols, vals) in _cases(rng):
            m, n, nnz = rowptr.size - 1, int(cols.max()) + 1, int(rowptr[-1])
            x, y0 = rng.standard_normal(n), rng.standard_normal(m)
            drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
            out = {}
            for alpha, beta in ((0.5, -2.0), (1.0, 0.0)):
                ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
                for mode in (0, rec):
                    hiplib.spmv_acc_reset_tunables()
                    for k, v in dict(pins, col16=mode, stream_plain=1).items():
                        assert hiplib.spmv_acc_set_tunable(k.encode(), v) == 0
                    dy = dev(torch, y0)
                    spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, dy, strategy=strategy)
                    torch.cuda.synchronize()
                    got = dy.cpu().numpy()
                    info = spmv_acc_amd.query_plan(drp, m)
                    assert oracle.scaled_error(got, ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (tag, strategy, mode, alpha, beta)
                    if info["last_kernel"] in ("rowblock", "flat_tile"):
                        # (1 = the record size the escape statistics choose, or none where even 64-int records would overflow: "long rows")
                        assert info["col16"] == mode if mode != 1 else info["col16"] in (0, 16, 32, 64), (tag, strategy, mode, info)
                    out[mode] = got
                    spmv_acc_amd.release_plans(drp)
                if strategy == "flat":
                    assert np.array_equal(out[0], out[rec]), (tag, "flat: same tile origins, same sums")
                else:
                    assert np.allclose(out[0], out[rec], rtol=0, atol=1e-12 * max(1.0, float(np.abs(ref).max()))), tag
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def test_record_size_follows_the_escape_statistics(torch_dev, oracle, hiplib):
    """The record size is the smallest of 16 / 32 / 64 ints that at most 1 % of the 256-non-zero chunks overflow (12 / 28 / 60 escapes); a matrix
    whose columns are random everywhere is not encoded at all, nor is one of fewer than 64 chunks.  (Size rules: tests/size_thresholds.py.)"""
    torch = torch_dev
    rng = np.random.default_rng(62)
    expect = [
        ("2 % far", synth.csr_from_row_lengths(rng.integers(20, 40, size=20000), 20000, rng, locality=300, far_fraction=0.02), 16),
        ("6 % far", synth.csr_from_row_lengths(rng.integers(20, 40, size=20000), 2_000_000, rng, locality=300, far_fraction=0.06), 32),
        ("15 % far", synth.csr_from_row_lengths(rng.integers(20, 40, size=20000), 2_000_000, rng, locality=300, far_fraction=0.15), 64),
        ("random columns", synth.csr_from_row_lengths(rng.integers(10, 30, size=20000), 3_000_000, rng, locality=1_400_000, far_fraction=0.5), 0),
        ("63 chunks", synth.csr_from_row_lengths(np.full(1008, 16), 1008, rng, locality=50, far_fraction=0.0), 0),
    ]
    try:
        hiplib.spmv_acc_reset_tunables()
        assert hiplib.spmv_acc_set_tunable(b"col16", 1) == 0
        for tag, (rowptr, cols, vals), want in expect:
            m, n, nnz = rowptr.size - 1, int(cols.max()) + 1, int(rowptr[-1])
            x, y0 = rng.standard_normal(n), rng.standard_normal(m)
            drp, dci, dv, dx, dy = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, dy, strategy="line_enhance")
            torch.cuda.synchronize()
            info = spmv_acc_amd.query_plan(drp, m)
            assert info["last_kernel"] == "rowblock" and info["col16"] == want, (tag, info)
            ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
            assert oracle.scaled_error(dy.cpu().numpy(), ref, 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, tag
            spmv_acc_amd.release_plans(drp)
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def test_timed_choice_is_kept_reported_and_bitwise_stable(torch_dev, oracle, hiplib):
    """Default (col16 = -1): spmv_acc_prepare builds the encoding, times it against colindex for the family that runs and settles; whatever it keeps,
    later calls are launches only, repeat bitwise, and spmv_acc_query_plan_col16 says which stream the kernel reads.  `deterministic` never uses it."""
    torch = torch_dev
    rng = np.random.default_rng(63)
    rowptr, cols, vals = synth.csr_from_row_lengths(rng.integers(18, 30, size=120000), 120000, rng, locality=400, far_fraction=0.02)
    m, n, nnz = rowptr.size - 1, int(cols.max()) + 1, int(rowptr[-1])
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
    ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
    try:
        for strategy in ("line_enhance", "flat", "adaptive"):
            hiplib.spmv_acc_reset_tunables()
            spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy=strategy)
            outs = []
            for _ in range(3):
                dy = dev(torch, y0)
                spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, dy, strategy=strategy)
                torch.cuda.synchronize()
                outs.append(dy.cpu().numpy())
            info = spmv_acc_amd.query_plan(drp, m)
            assert info["settled"] and info["col16"] in (0, 16), (strategy, info)
            assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[1], outs[2]), strategy
            assert oracle.scaled_error(outs[0], ref, 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, strategy
            spmv_acc_amd.release_plans(drp)
        hiplib.spmv_acc_reset_tunables()
        assert hiplib.spmv_acc_set_tunable(b"deterministic", 1) == 0
        dy = dev(torch, y0)
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, dy, strategy="line_enhance")
        torch.cuda.synchronize()
        assert spmv_acc_amd.query_plan(drp, m)["col16"] == 0
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


@pytest.mark.parametrize("strategy", ["line_enhance", "flat"])
def test_row_shards_without_rebasing(torch_dev, oracle, hiplib, strategy):
    """Un-rebased row sub-ranges (rowptr + r0, whole colindex / value arrays): the encoding is built for the view's own chunks -- from the flat tile that
    holds the view's first non-zero on -- and indexed by absolute non-zero index.  (Round 6's first form started at the view's first CHUNK: a flat
    tile, whose origin is a multiple of 2048, read records in front of the table -- a memory fault on the GPU box; this test holds that case.)"""
    torch = torch_dev
    rng = np.random.default_rng(64)
    rowptr, cols, vals = synth.csr_from_row_lengths(rng.integers(10, 60, size=30000), 30000, rng, locality=300, far_fraction=0.03)
    x, y0 = rng.standard_normal(30000), rng.standard_normal(30000)
    ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
    try:
        hiplib.spmv_acc_reset_tunables()
        for k, v in (("col16", 1), ("flat_rowblock", 0)):
            assert hiplib.spmv_acc_set_tunable(k.encode(), v) == 0
        for r0, r1 in ((12345, 27001), (1, 30000), (3001, 7777), (29000, 30000)):
            assert int(rowptr[r0]) % 2048 != 0
            dy = dev(torch, y0)
            spmv_acc_amd.csr_spmv(1.0, 1.0, r1 - r0, 30000, int(rowptr[r1]), drp[r0:], dci, dv, dx, dy[r0:], strategy=strategy)
            torch.cuda.synchronize()
            got = dy.cpu().numpy()
            assert np.array_equal(got[:r0], y0[:r0]) and np.array_equal(got[r1:], y0[r1:]), (r0, r1, "wrote outside the shard")
            assert np.max(np.abs(got[r0:r1] - ref[r0:r1])) <= 1e-11, (r0, r1)
            chunks = (int(rowptr[r1]) + 255) // 256 - int(rowptr[r0]) // 2048 * 8
            assert spmv_acc_amd.query_plan(drp[r0:], r1 - r0)["col16"] == (16 if chunks >= 64 else 0), (r0, r1, chunks)
            spmv_acc_amd.release_plans(drp[r0:])
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def test_colindex_edited_in_place_is_noticed(torch_dev, oracle, hiplib):
    """The hazard the encoding adds: the kernel no longer reads colindex, so a caller who rewrites the column indices in place (same rowptr, no
    spmv_acc_release_plans) would silently keep the old columns.  Every launch re-checks 64 samples of colindex against the plan's copies (as for
    rowptr): a wholesale rewrite raises the plan's stale flag, the library reports it and rebuilds, and results match the oracle again.  Editing
    VALUES in place stays free: they are streamed from the caller's array."""
    torch = torch_dev
    rng = np.random.default_rng(65)
    rowptr, cols, vals = synth.csr_from_row_lengths(rng.integers(20, 40, size=20000), 300000, rng, locality=300, far_fraction=0.02)
    m, n, nnz = rowptr.size - 1, 300000, int(rowptr[-1])
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))

    def spmv():
        dy = dev(torch, y0)
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, dy, strategy="line_enhance")
        torch.cuda.synchronize()
        return dy.cpu().numpy()

    try:
        hiplib.spmv_acc_reset_tunables()
        assert hiplib.spmv_acc_set_tunable(b"col16", 1) == 0
        hiplib.spmv_acc_clear_error()
        got = spmv()
        assert spmv_acc_amd.query_plan(drp, m)["col16"] == 16
        assert oracle.scaled_error(got, oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0), 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL
        vals2 = vals * -0.75
        dv.copy_(dev(torch, vals2))
        got = spmv()
        assert hiplib.spmv_acc_last_error() == 0
        assert oracle.scaled_error(got, oracle.host_spmv(1.0, 1.0, rowptr, cols, vals2, x, y0), 1.0, 1.0, rowptr, cols, vals2, x, y0) <= SCALED_TOL
        # every column mirrored inside its row's band: same rowptr, every colindex entry (almost) different
        cols2 = np.concatenate([np.sort((n - 1 - cols[rowptr[i]:rowptr[i + 1]])) for i in range(m)]).astype(np.int32)
        dci.copy_(dev(torch, cols2))
        torch.cuda.synchronize()
        reported = False
        try:
            spmv()  # the stale plan's y is not to be trusted
        except spmv_acc_amd.SpmvAccError:
            reported = True
        reported = reported or hiplib.spmv_acc_last_error() != 0
        assert reported, "the colindex samples did not raise the stale flag"
        hiplib.spmv_acc_clear_error()
        try:
            got = spmv()
        except spmv_acc_amd.SpmvAccError:  # (the report may surface on this call instead: the one after it runs on the rebuilt plan)
            hiplib.spmv_acc_clear_error()
            got = spmv()
        assert oracle.scaled_error(got, oracle.host_spmv(1.0, 1.0, rowptr, cols2, vals2, x, y0), 1.0, 1.0, rowptr, cols2, vals2, x, y0) <= SCALED_TOL
    finally:
        hiplib.spmv_acc_clear_error()
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()
