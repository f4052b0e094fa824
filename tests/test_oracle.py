"""CPU suite: the oracle (oracle/spmv_oracle.c) against golden vectors, an independent implementation
(scipy) and closed-form cases; the oracle's analysis against the compiled reference (oracle/_ref)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from spmv_acc_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_host_spmv_identity_and_diagonal(oracle):
    m = 257
    rowptr = np.arange(m + 1, dtype=np.int32)
    cols = np.arange(m, dtype=np.int32)
    d = np.linspace(-2, 3, m)
    x = np.linspace(1, 2, m)
    y0 = np.full(m, 0.25)
    y = oracle.host_spmv(2.0, -4.0, rowptr, cols, d, x, y0)
    assert np.array_equal(y, 2.0 * (d * x) + (-4.0) * y0)  # one product per row: exact


def test_host_spmv_left_to_right_order(oracle):
    # 1e16 + 1 - 1e16 depends on the summation order: left-to-right gives 0, not 1
    rowptr = np.array([0, 3], dtype=np.int32)
    cols = np.array([0, 1, 2], dtype=np.int32)
    vals = np.array([1e16, 1.0, -1e16])
    y = oracle.host_spmv(1.0, 0.0, rowptr, cols, vals, np.ones(3), np.zeros(1))
    assert y[0] == 0.0


def test_host_spmv_golden_and_scipy(oracle):
    g = np.load(os.path.join(GOLD, "spmv_cases.npz"))
    for name in g["names"]:
        rowptr, cols, vals = g[f"{name}__rowptr"], g[f"{name}__cols"], g[f"{name}__vals"]
        x, y0 = g[f"{name}__x"], g[f"{name}__y0"]
        m = rowptr.size - 1
        A = sp.csr_matrix((vals, cols, rowptr), shape=(m, x.size))
        for k, (a, b) in enumerate(g["alpha_beta"]):
            y = oracle.host_spmv(a, b, rowptr, cols, vals, x, y0)
            assert np.array_equal(y, g[f"{name}__out{k}"]), (name, a, b)  # bit-exact vs the committed vector
            ys = a * (A @ x) + b * y0
            assert oracle.scaled_error(y, ys, a, b, rowptr, cols, vals, x, y0) < 1e-13


def test_host_spmv_omp_bit_identical(oracle):
    rowptr, cols, vals = synth.random_csr(5000, 5000, 7, seed=5, kind="powerlaw")
    rng = np.random.default_rng(1)
    x, y0 = rng.standard_normal(5000), rng.standard_normal(5000)
    ref = oracle.host_spmv(1.5, 0.5, rowptr, cols, vals, x, y0)
    for t in (1, 2, 3):
        y = y0.copy()
        oracle.host_spmv_omp(1.5, 0.5, rowptr, cols, vals, x, y, t)
        assert np.array_equal(y, ref)


def test_host_spmv_bench_form_is_the_same_arithmetic(oracle):
    """bench.py's cpu_baseline times oracle_host_spmv_bench (arrays re-placed by first touch, y restored before every run): per row
    the arithmetic of cli/verification.cpp:56-66 still, so y is bitwise the sequential form's -- for any thread count, a rectangular
    matrix (x is placed by column ranges n/m per row), empty rows and more threads than rows."""
    rng = np.random.default_rng(7)
    for m, n, avg, kind in ((5000, 3000, 7, "powerlaw"), (300, 9000, 3, "uniform"), (5, 40, 2, "uniform")):
        rowptr, cols, vals = synth.random_csr(m, n, avg, seed=m, kind=kind)
        x, y0 = rng.standard_normal(n), rng.standard_normal(m)
        ref = oracle.host_spmv(0.5, -2.0, rowptr, cols, vals, x, y0)
        for t in (1, 3, 8):
            secs, y = oracle.host_spmv_bench(0.5, -2.0, rowptr, cols, vals, x, y0, t, 3)
            assert np.array_equal(y, ref), (m, n, t)
            assert secs.shape == (3,) and (secs >= 0).all()
    assert oracle.stream_triad_gbs(1 << 16, 2, 2) > 0


def test_verify_thresholds(oracle):
    hy = np.array([1.0, 2.0, 1e-13, 0.0, 5.0])
    assert oracle.verify(hy.copy(), hy) == -1  # 0/0 = NaN passes, as in the reference
    dy = hy.copy()
    dy[1] = 2.0 * (1 + 2e-7)
    assert oracle.verify(dy, hy) == 1
    dy = hy.copy()
    dy[1] = 2.0 * (1 + 5e-8)
    assert oracle.verify(dy, hy) == -1
    dy = hy.copy()
    dy[3] = 1e-20  # hy == 0, dy != 0 -> inf -> fails in the CLI verdict
    assert oracle.verify(dy, hy) == 3
    # benchmark verdict: |hy| <= 1e-12 uses the absolute 1e-14 gate
    me, first, cnt = oracle.verify_y(dy, hy)
    assert (first, cnt) == (-1, 0) and me == 1e-20
    dy[2] = 1e-13 + 2e-14
    me, first, cnt = oracle.verify_y(dy, hy)
    assert (first, cnt) == (2, 1)


def test_rand_grid(oracle):
    L = oracle.lib()
    L.oracle_srand(1)
    x = np.zeros(1000)
    L.oracle_rand_vector(1000, x.ctypes.data_as(oracle._dp))
    grid = -1.0 + 2.0 * np.arange(100) / 101.0
    assert np.all(np.isin(x, grid)) and x.min() >= -1.0 and x.max() <= 0.9604


def test_metric_formulas(oracle):
    L = oracle.lib()
    rows, nnz = 8_217_820, 40_451_632
    assert L.oracle_ref_mem_bytes(rows, nnz) == 8 * (2 * rows + nnz) + 4 * (rows + 1 + nnz)
    assert L.oracle_ref_mem_bytes(rows, nnz) == synth.reference_bytes(rows, nnz)
    assert abs(L.oracle_ref_gflops(nnz, 100.0) - 2 * nnz / 100.0 / 1e3) < 1e-9
    assert abs(L.oracle_ref_gibps(rows, nnz, 100.0) - synth.reference_bytes(rows, nnz) / 2**30 / 100e-6) < 1e-6


# ---- break points (device form of the preprocessing pass) -------------------------------------------
def _bp_closed_form(rowptr, stride):
    """Independent statement of flat_imp.inl:108-131's result (used to cross-check the restatement)."""
    m = rowptr.size - 1
    nnz = int(rowptr[m])
    n = nnz // stride + (1 if nnz % stride else 0) + 1
    bp = np.zeros(n, dtype=np.int32)
    for j in range(1, n):
        t = j * stride
        if t > nnz:
            continue
        p = int(np.searchsorted(rowptr, t, side="left"))
        bp[j] = p if rowptr[p] == t else p - 1
    return bp


def test_break_points_golden_and_closed_form(oracle):
    g = np.load(os.path.join(GOLD, "breakpoint_cases.npz"))
    for name in g["names"]:
        rp = g[f"{name}__rowptr"]
        for s in g["strides"]:
            bp = oracle.break_points(rp, int(s))
            assert np.array_equal(bp, g[f"{name}__{s}"]), (name, s)
            assert np.array_equal(bp, _bp_closed_form(rp, int(s))), (name, s)


def test_break_points_random(oracle):
    rng = np.random.default_rng(11)
    for trial in range(60):
        m = int(rng.integers(1, 400))
        lens = rng.integers(0, [3, 40, 3000][trial % 3], m)
        rp = np.zeros(m + 1, dtype=np.int32)
        np.cumsum(lens, out=rp[1:])
        for s in (64, 1024):
            assert np.array_equal(oracle.break_points(rp, s), _bp_closed_form(rp, s))


def test_break_points_v2_matches_v1_without_empty_rows(oracle):
    rng = np.random.default_rng(12)
    lens = rng.integers(1, 50, 500)  # v2's known defect needs empty leading rows
    rp = np.zeros(501, dtype=np.int32)
    np.cumsum(lens, out=rp[1:])
    v1 = oracle.break_points(rp, 256)
    v2 = oracle.break_points(rp, 256, v2=True)
    nblk = v1.size - 1
    # v2 labels block j with the row holding its first nnz; v1 agrees except where a block starts
    # exactly on a row boundary (both then name the starting row) -- identical for j < nblk
    assert np.array_equal(v1[:nblk], v2[:nblk])


# ---- adaptive-plus analysis (host form of the preprocessing pass) ---------------------------------------
def test_analysis_golden_from_reference(oracle):
    g = np.load(os.path.join(GOLD, "analysis_cases.npz"))
    for name in g["names"]:
        rp = g[f"{name}__rowptr"]
        for k, (threads, vec, min_nnz) in enumerate(g["params"]):
            blocks, bp, fbr = oracle.adaptive_plus_analyze(rp, int(min_nnz), int(threads), int(vec))
            assert np.array_equal(bp, g[f"{name}__{k}__bp"]), (name, k)
            assert np.array_equal(fbr, g[f"{name}__{k}__fbr"]), (name, k)
            assert blocks == bp.size - 1


def test_analysis_against_compiled_reference(oracle):
    if oracle.ref() is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    rng = np.random.default_rng(99)
    for trial in range(120):
        m = int(rng.integers(1, 2500))
        kind = trial % 4
        if kind == 0:
            lens = rng.integers(0, 12, m)
        elif kind == 1:
            lens = np.minimum((rng.pareto(1.2, m) * 3).astype(np.int64), 20000)
        elif kind == 2:
            lens = rng.integers(0, 3, m)
            lens[rng.integers(0, m, 3)] = rng.integers(2000, 30000, 3)
        else:
            lens = rng.integers(0, 700, m)
        rp = np.zeros(m + 1, dtype=np.int32)
        np.cumsum(lens, out=rp[1:])
        for threads, vec, min_nnz in ((512, 1, 2048), (512, 4, 2048), (256, 16, 1024), (1024, 64, 4096)):
            a = oracle.adaptive_plus_analyze(rp, min_nnz, threads, vec)
            b = oracle.ref_adaptive_plus_analyze(rp, min_nnz, threads, vec)
            assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_strategy_pickers(oracle):
    L = oracle.lib()
    # Hardesty3's statistics (examples/large-data-set-batch.sh:39-40): 40451632 / 8217820 = 4 -> adaptive line
    m, nnz = 8_217_820, 40_451_632
    rp = np.linspace(0, nnz, m + 1).astype(np.int32)
    assert oracle.adaptive_pick(rp) == 2
    v, rn = np.zeros(1, np.int32), np.zeros(1, np.int32)
    L.oracle_adaptive_line_params(m, nnz, v.ctypes.data_as(oracle._ip), rn.ctypes.data_as(oracle._ip))
    assert (int(v[0]), int(rn[0])) == (2, 102)
    # large, long rows -> flat; small -> line-enhance
    m, nnz = 914_898, 28_191_660
    assert oracle.adaptive_pick(np.linspace(0, nnz, m + 1).astype(np.int32)) == 4
    m, nnz = 268_096, 9_378_286
    assert oracle.adaptive_pick(np.linspace(0, nnz, m + 1).astype(np.int32)) == 3
    # unbalanced halves -> vector-row split
    rp = np.concatenate([np.arange(0, 500), 499 + 40 * np.arange(1, 502)]).astype(np.int32)
    assert oracle.adaptive_pick(rp) == 1
    assert L.oracle_adaptive_vec_row_bp(100, 900) == 2 and L.oracle_adaptive_vec_row_bp(1, 10**6) == 1
    assert L.oracle_adaptive_plus_vec(8_217_820, 40_451_632) == 2
