"""GPU suite (-m gpu): plans and their guards (stale plans detected and rebuilt, the ten-argument entry with a changed nnz, the wrapper's argument
checks, the library stream following torch's), the one-rank RCCL ordering of the row-sharded step, and kernel invariants that must not leak into a
sum (walking direction, row digest edge cases, the segmented-scan reduction, gather hints).  (Until round 6: test_gpu_round2.py.)"""
import os
import socket

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


def _two_matrices_same_shape_and_nnz(m, n, seed):
    """A and B: same m, n and nnz (B's row lengths are A's in reverse order), different structure and values."""
    rng = np.random.default_rng(seed)
    lens = np.minimum((rng.pareto(1.3, size=m) * 6).astype(np.int64), 3000) + rng.integers(0, 4, size=m)
    lens[: m // 3] += 9  # make the reversal change every rowptr entry by a lot
    A = synth.csr_from_row_lengths(lens, n, rng)
    B = synth.csr_from_row_lengths(lens[::-1].copy(), n, rng)
    assert A[0][-1] == B[0][-1] and not np.array_equal(A[0], B[0])
    return A, B


@pytest.mark.parametrize("strat", ["flat", "adaptive_plus", "adaptive", "line_enhance", "vector_row"])
def test_stale_plan_is_detected_and_rebuilt(torch_dev, oracle, hiplib, strat):
    """A caller rebuilds a DIFFERENT matrix with equal m, n, nnz in the SAME buffers and does not call
    spmv_acc_release_plans (the reference recomputes its preprocessing per call, flat.cpp:39-44, so its callers never
    announce a change).  The first SpMV on the new matrix runs with the old plan and raises the plan's sticky flag;
    spmv_acc_last_error() / the next call report SPMV_ACC_ERR_BAD_ARGUMENT, the plan is rebuilt, and from then on results
    match the oracle again.  Editing VALUES in place (same structure) must not trip the guard."""
    torch = torch_dev
    m = n = 30000
    (rpA, ciA, vA), (rpB, ciB, vB) = _two_matrices_same_shape_and_nnz(m, n, seed=11)
    nnz = int(rpA[-1])
    rng = np.random.default_rng(12)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx = dev(torch, rpA), dev(torch, ciA), dev(torch, vA), dev(torch, x)

    def spmv():
        dy = dev(torch, y0)
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, dy, strategy=strat)
        torch.cuda.synchronize()
        return dy.cpu().numpy()

    try:
        hiplib.spmv_acc_clear_error()
        got = spmv()
        assert oracle.scaled_error(got, oracle.host_spmv(1.0, 1.0, rpA, ciA, vA, x, y0), 1.0, 1.0, rpA, ciA, vA, x, y0) <= SCALED_TOL
        plans = hiplib.spmv_acc_cached_plans()
        # values edited in place: same plan, no complaint, new result
        vA2 = vA * 1.5
        dv.copy_(dev(torch, vA2))
        got = spmv()
        assert hiplib.spmv_acc_last_error() == 0 and hiplib.spmv_acc_cached_plans() == plans
        assert oracle.scaled_error(got, oracle.host_spmv(1.0, 1.0, rpA, ciA, vA2, x, y0), 1.0, 1.0, rpA, ciA, vA2, x, y0) <= SCALED_TOL
        # the other matrix, same buffers, no release
        drp.copy_(dev(torch, rpB))
        dci.copy_(dev(torch, ciB))
        dv.copy_(dev(torch, vB))
        torch.cuda.synchronize()
        reported = False
        try:
            spmv()  # stale plan: this y is not to be trusted (the kernel may already have raised the flag when the
        except spmv_acc_amd.SpmvAccError as ex:  # wrapper looks at the error slot)
            reported = "changed" in str(ex)
        if not reported:
            assert hiplib.spmv_acc_last_error() == 2  # SPMV_ACC_ERR_BAD_ARGUMENT, raised by the kernel's guard
            assert b"changed" in hiplib.spmv_acc_last_error_string()
        hiplib.spmv_acc_clear_error()
        refB = oracle.host_spmv(1.0, 1.0, rpB, ciB, vB, x, y0)
        for _ in range(2):  # fresh plan: correct, and no further complaint
            got = spmv()
            assert hiplib.spmv_acc_last_error() == 0
            assert oracle.scaled_error(got, refB, 1.0, 1.0, rpB, ciB, vB, x, y0) <= SCALED_TOL, strat
        # second scenario: nobody asks for the error -- the next call on the matrix reports it and still computes correctly
        drp.copy_(dev(torch, rpA))
        dci.copy_(dev(torch, ciA))
        dv.copy_(dev(torch, vA))
        torch.cuda.synchronize()
        dy = dev(torch, y0)
        hiplib.spmv_acc_csr_spmv_strategy(spmv_acc_amd.strategy_id(strat), 0, 1.0, 1.0, m, n, nnz, None, drp.data_ptr(),
                                          dci.data_ptr(), dv.data_ptr(), dx.data_ptr(), dy.data_ptr())
        torch.cuda.synchronize()  # (raw C call: nothing looked at the error slot)
        dy = dev(torch, y0)
        with pytest.raises(spmv_acc_amd.SpmvAccError, match="changed"):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, dy, strategy=strat)
        torch.cuda.synchronize()
        refA = oracle.host_spmv(1.0, 1.0, rpA, ciA, vA, x, y0)
        assert oracle.scaled_error(dy.cpu().numpy(), refA, 1.0, 1.0, rpA, ciA, vA, x, y0) <= SCALED_TOL  # rebuilt before it ran
    finally:
        hiplib.spmv_acc_clear_error()
        spmv_acc_amd.release_plans(drp)


def test_sparse_spmv_ten_args_with_changed_nnz(torch_dev, oracle, hiplib):
    """The ten-argument entry never receives nnz: a matrix with ANOTHER nnz behind the same pointers (same m, n) is caught
    by the guard as well (rowptr[m] is one of the samples)."""
    torch = torch_dev
    m = n = 20000
    rpA, ciA, vA = synth.random_csr(m, n, 8, seed=3, kind="uniform")
    rpB, ciB, vB = synth.random_csr(m, n, 5, seed=4, kind="uniform")
    assert rpA[-1] > rpB[-1]
    rng = np.random.default_rng(5)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    cap = int(rpA[-1])
    drp, dx = dev(torch, rpA), dev(torch, x)
    dci = torch.zeros(cap, dtype=torch.int32, device="cuda")
    dv = torch.zeros(cap, dtype=torch.float64, device="cuda")
    dci.copy_(dev(torch, ciA))
    dv.copy_(dev(torch, vA))
    try:
        hiplib.spmv_acc_clear_error()
        dy = dev(torch, y0)
        spmv_acc_amd.sparse_spmv(0, 1.0, 1.0, m, n, drp, dci, dv, dx, dy)
        torch.cuda.synchronize()
        assert oracle.verify(dy.cpu().numpy(), oracle.host_spmv(1.0, 1.0, rpA, ciA, vA, x, y0)) == -1
        drp.copy_(dev(torch, rpB))
        dci[: ciB.size].copy_(dev(torch, ciB))
        dv[: vB.size].copy_(dev(torch, vB))
        torch.cuda.synchronize()
        try:
            spmv_acc_amd.sparse_spmv(0, 1.0, 1.0, m, n, drp, dci, dv, dx, dev(torch, y0))
        except spmv_acc_amd.SpmvAccError:
            pass
        torch.cuda.synchronize()
        hiplib.spmv_acc_last_error()  # drops the stale plan if the wrapper had not seen the flag yet
        hiplib.spmv_acc_clear_error()
        dy = dev(torch, y0)
        spmv_acc_amd.sparse_spmv(0, 1.0, 1.0, m, n, drp, dci, dv, dx, dy)
        torch.cuda.synchronize()
        assert oracle.verify(dy.cpu().numpy(), oracle.host_spmv(1.0, 1.0, rpB, ciB, vB, x, y0)) == -1
        assert spmv_acc_amd.query_plan(drp, m)["nnz"] == int(rpB[-1])
    finally:
        hiplib.spmv_acc_clear_error()
        spmv_acc_amd.release_plans(drp)


def test_wrapper_refuses_what_the_kernels_would_misread(torch_dev):
    torch = torch_dev
    rowptr, cols, vals = synth.random_csr(500, 500, 5, seed=1)
    m = n = 500
    nnz = int(rowptr[-1])
    drp, dci, dv = dev(torch, rowptr), dev(torch, cols), dev(torch, vals)
    dx = torch.ones(n, dtype=torch.float64, device="cuda")
    dy = torch.zeros(m, dtype=torch.float64, device="cuda")
    bad = [
        dict(rowptr=drp.long()),                                   # torch's default integer type
        dict(value=dv.float()),                                    # fp32 values
        dict(x=torch.ones(2 * n, dtype=torch.float64, device="cuda")[::2]),  # strided view
        dict(y=torch.zeros(m - 1, dtype=torch.float64, device="cuda")),      # too short
        dict(colindex=dci[: nnz - 1]),
        dict(x=torch.ones(n, dtype=torch.float64)),                # host tensor
    ]
    for override in bad:
        args = dict(rowptr=drp, colindex=dci, value=dv, x=dx, y=dy)
        args.update(override)
        with pytest.raises(spmv_acc_amd.SpmvAccError):
            spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, nnz, args["rowptr"], args["colindex"], args["value"], args["x"], args["y"])
    spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, nnz, drp, dci, dv, dx, dy)  # and the good call goes through
    torch.cuda.synchronize()
    spmv_acc_amd.release_plans(drp)


def test_library_follows_torch_current_stream(torch_dev, oracle):
    """The wrappers point the library stream at torch's current stream: work queued on a side stream before the call
    (here: a long sleep, then the write of x) is seen by the SpMV without any host synchronisation."""
    torch = torch_dev
    rowptr, cols, vals = synth.random_csr(20000, 20000, 7, seed=9)
    m = n = 20000
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx_src, dy = dev(torch, rowptr), dev(torch, cols), dev(torch, vals), dev(torch, x), dev(torch, y0)
    spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx_src, strategy="adaptive")
    dx = torch.zeros(n, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        torch.cuda._sleep(100_000_000)  # tens of milliseconds
        dx.copy_(dx_src)
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, dy, strategy="adaptive")
    side.synchronize()
    ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
    assert oracle.scaled_error(dy.cpu().numpy(), ref, 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL
    spmv_acc_amd.load_library().spmv_acc_set_stream(None)
    spmv_acc_amd.release_plans(drp)


def test_row_sharded_event_ordering_on_one_rank_rccl(torch_dev, oracle):
    """RowShardedSpmv on a ONE-rank RCCL communicator (all this box allows): the local SpMV runs on the engine's own
    non-NULL stream, the allgather is issued behind the event recorded after it, and the in-place iteration
    (beta != 0, no y_prev) follows the serial recurrence.  The compute stream is held back by a long sleep before a
    step: an exchange that did not wait for the SpMV's event would gather the old buffer."""
    torch = torch_dev
    import torch.distributed as dist

    from spmv_acc_amd.dist import RowShardedSpmv

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        m = n = 200_000
        rowptr, cols, vals = synth.random_csr(m, n, 6, seed=17)
        nnz = int(rowptr[-1])
        rng = np.random.default_rng(4)
        x, y0 = rng.standard_normal(n), rng.standard_normal(m)
        drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
        bounds = np.array([0, m], dtype=np.int64)
        for exchange in ("allgather", "p2p"):
            eng = RowShardedSpmv(0, 1, bounds, drp, dci, dv, n, torch.device("cuda", 0), strategy="adaptive",
                                 always_collective=True, exchange=exchange, own_stream=True)
            assert eng.compute_stream is not None and eng.compute_stream.cuda_stream != 0  # explicit non-NULL stream
            assert eng.compute_stream.cuda_stream != torch.cuda.current_stream().cuda_stream
            eng.prepare(0.5, dx)  # the plan, every per-matrix timing settled (a later step that still tuned would synchronise)
            eng.set_y(dy0)
            eng.step(0.25, 0.5, dx)
            eng.wait()
            torch.cuda.synchronize()
            with torch.cuda.stream(eng.compute_stream):
                torch.cuda._sleep(200_000_000)  # the next SpMV sits behind this
            eng.step(0.25, 0.5, dx, overlap=True)
            assert eng.exchange_issued_after_spmv and not eng.spmv_done.query()  # SpMV still pending: the exchange must wait for it
            eng.step(0.25, 0.5, dx, overlap=True)
            got = eng.gathered().cpu().numpy()
            assert eng.spmv_done.query()
            y = y0
            for _ in range(3):
                y = oracle.host_spmv(0.25, 0.5, rowptr, cols, vals, x, y)
            scale_ok = oracle.scaled_error(got, y, 0.25, 0.5, rowptr, cols, vals, x, y0)
            assert scale_ok <= 1e-11, (exchange, scale_ok)
            # the default (round 3): the local SpMV on the CURRENT stream, ordered with the exchange by stream order alone; held
            # back the same way, same recurrence
            eng = RowShardedSpmv(0, 1, bounds, drp, dci, dv, n, torch.device("cuda", 0), strategy="adaptive",
                                 always_collective=True, exchange=exchange)
            eng.set_y(dy0)
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                torch.cuda._sleep(200_000_000)
                for _ in range(3):
                    eng.step(0.25, 0.5, dx, overlap=True)
                got2 = eng.gathered().cpu().numpy()
            assert oracle.scaled_error(got2, y, 0.25, 0.5, rowptr, cols, vals, x, y0) <= 1e-11, (exchange, "current-stream form")
        spmv_acc_amd.release_plans(drp)
    finally:
        dist.destroy_process_group()
        spmv_acc_amd.load_library().spmv_acc_set_stream(None)


def test_opt_in_col16_encoding_matches_plain_flat(torch_dev, oracle, hiplib):
    """Tunable col16 (off by default): flat reads the plan's 16-bit column encoding (per-256-non-zero base + escape list)
    instead of colindex.  Same products in the same order -> bit-identical to flat without it (same tile size and cut-row
    form), on local columns (few escapes), on far columns (10 %), on random columns (almost every entry escapes), on every
    residue of nnz mod 4 (the ragged last group reads colindex as usual) and on a row shard that is not rebased."""
    torch = torch_dev
    cases = []
    rng = np.random.default_rng(21)
    cases.append(("fem-like", synth.csr_from_row_lengths(rng.integers(20, 40, size=20000), 20000, rng, locality=300, far_fraction=0.02)))
    cases.append(("far 10 %", synth.csr_from_row_lengths(rng.integers(3, 8, size=60000), 55000, rng, locality=64, far_fraction=0.10)))
    cases.append(("random columns", synth.csr_from_row_lengths(rng.integers(0, 30, size=8000), 3_000_000, rng, locality=1_400_000, far_fraction=0.5)))
    cases.append(("long rows", synth.csr_from_row_lengths(rng.integers(900, 5000, size=300), 200000, rng, locality=90000, far_fraction=0.0)))
    for r in range(4):
        lens = rng.integers(1, 9, size=3000)
        lens[-1] += (r - int(lens.sum())) % 4  # nnz mod 4 == r
        cases.append((f"nnz mod 4 = {r}", synth.csr_from_row_lengths(lens, 3000, rng)))
    try:
        for tag, (rowptr, cols, vals) in cases:
            m, n, nnz = rowptr.size - 1, int(cols.max()) + 1 if cols.size else 1, int(rowptr[-1])
            x, y0 = rng.standard_normal(n), rng.standard_normal(m)
            drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
            ref = oracle.host_spmv(0.5, -2.0, rowptr, cols, vals, x, y0)
            out = {}
            for mode in (0, 1):
                for finish in (0, 1):
                    hiplib.spmv_acc_reset_tunables()
                    for k, val in (("col16", mode), ("flat_npt", 8), ("flat_finish", finish), ("flat_early", 0), ("stream_plain", 1),
                                   ("flat_rowblock", 0)):  # (round 3: small balanced grids may otherwise run the row-block kernel)
                        assert hiplib.spmv_acc_set_tunable(k.encode(), val) == 0
                    dy = dev(torch, y0)
                    spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, dy, strategy="flat")
                    torch.cuda.synchronize()
                    out[(mode, finish)] = dy.cpu().numpy()
                    assert oracle.scaled_error(out[(mode, finish)], ref, 0.5, -2.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, (tag, mode, finish)
                    spmv_acc_amd.release_plans(drp)
            for finish in (0, 1):
                assert np.array_equal(out[(0, finish)], out[(1, finish)]), (tag, finish)
        # a row shard handed over without rebasing (rowptr[0] > 0, offsets into the whole colindex / value arrays)
        rowptr, cols, vals = synth.csr_from_row_lengths(rng.integers(10, 30, size=9000), 9000, rng, locality=200, far_fraction=0.05)
        x, y0 = rng.standard_normal(9000), rng.standard_normal(9000)
        r0, r1 = 3001, 7777
        drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
        hiplib.spmv_acc_reset_tunables()
        assert hiplib.spmv_acc_set_tunable(b"col16", 1) == 0
        dy = dev(torch, y0[r0:r1])
        spmv_acc_amd.csr_spmv(1.0, 1.0, r1 - r0, 9000, int(rowptr[r1]), drp[r0: r1 + 1], dci, dv, dx, dy, strategy="flat")
        torch.cuda.synchronize()
        ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)[r0:r1]
        assert np.max(np.abs(dy.cpu().numpy() - ref)) <= 1e-11
        spmv_acc_amd.release_plans(drp[r0: r1 + 1])
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def test_c_abi_sharded_spmv_on_a_one_rank_communicator(torch_dev, oracle, hiplib):
    """spmv_acc_sharded_spmv: one rank's step of the row-sharded SpMV for C consumers -- local SpMV + ONE ncclAllGather of the
    padded y slices on the caller's communicator, RCCL resolved at run time from the copy the process already uses (here: the
    one bundled with torch, driven through ctypes exactly as a C caller would drive it).  One rank is all this box allows;
    the collective still runs through RCCL."""
    import ctypes

    torch = torch_dev
    rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"))

    class UniqueId(ctypes.Structure):
        _fields_ = [("internal", ctypes.c_char * 128)]

    torch.cuda.set_device(0)
    torch.zeros(1, device="cuda")  # the device is initialised before RCCL looks at it
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        m, n, pad = 50_000, 60_000, 50_048
        rowptr, cols, vals = synth.random_csr(m, n, 9, seed=5, kind="uniform")
        nnz = int(rowptr[-1])
        rng = np.random.default_rng(6)
        x, y0 = rng.standard_normal(n), rng.standard_normal(m)
        drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
        hiplib.spmv_acc_set_stream(None)
        for strat, (alpha, beta) in (("adaptive", (1.0, 1.0)), ("flat", (0.5, -2.0)), ("line_enhance", (1.0, 0.0))):
            y_local = torch.zeros(pad, dtype=torch.float64, device="cuda")
            y_local[:m] = dev(torch, y0)
            y_full = torch.full((pad,), float("nan"), dtype=torch.float64, device="cuda")
            torch.cuda.synchronize()
            rc = hiplib.spmv_acc_sharded_spmv(comm, spmv_acc_amd.strategy_id(strat), alpha, beta, m, pad, n, nnz, None,
                                              drp.data_ptr(), dci.data_ptr(), dv.data_ptr(), dx.data_ptr(), y_local.data_ptr(),
                                              y_full.data_ptr())
            assert rc == 0, hiplib.spmv_acc_last_error_string()
            torch.cuda.synchronize()
            got = y_full.cpu().numpy()
            ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
            assert oracle.scaled_error(got[:m], ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, strat
            assert np.all(got[m:] == 0.0)  # the padding travelled too
        # bad arguments are refused before anything is launched
        assert hiplib.spmv_acc_sharded_spmv(None, 1, 1.0, 0.0, m, pad, n, nnz, None, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(),
                                            dx.data_ptr(), y_local.data_ptr(), y_full.data_ptr()) == 2
        assert hiplib.spmv_acc_sharded_spmv(comm, 1, 1.0, 0.0, m, m - 1, n, nnz, None, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(),
                                            dx.data_ptr(), y_local.data_ptr(), y_full.data_ptr()) == 2
        hiplib.spmv_acc_clear_error()
        spmv_acc_amd.release_plans(drp)
    finally:
        rccl.ncclCommDestroy(comm)


def test_zigzag_direction_never_enters_a_sum(torch_dev, oracle, hiplib):
    """Consecutive SpMVs on a plan walk the grid in alternating directions (tunable zigzag, on by default): the direction is
    a matter of cache reuse only.  Four consecutive calls (forward, backward, forward, backward) of every tile-kernel family
    give bit-identical y, identical to the one-direction form, and match the oracle; block counts that are not a multiple of
    the eight XCDs and single-block grids included."""
    torch = torch_dev
    rng = np.random.default_rng(33)
    for m, avg in ((50_003, 9), (9, 3), (1, 5), (2051, 40), (70_000, 2)):
        rowptr, cols, vals = synth.csr_from_row_lengths(rng.integers(0, 2 * avg + 1, size=m), max(m, 64), rng)
        n, nnz = max(m, 64), int(rowptr[-1])
        x, y0 = rng.standard_normal(n), rng.standard_normal(m)
        drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
        ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
        for strat in ("line_enhance", "flat", "adaptive_plus", "vector_row"):
            outs = []
            for zz in (1, 0):
                hiplib.spmv_acc_reset_tunables()
                assert hiplib.spmv_acc_set_tunable(b"zigzag", zz) == 0
                for k, val in (("flat_finish", 0), ("stream_plain", 0), ("plus_min_nnz", 1024), ("flat_npt", 8)):  # one summation order
                    assert hiplib.spmv_acc_set_tunable(k.encode(), val) == 0
                spmv_acc_amd.release_plans(drp)
                for _ in range(4):
                    dy = dev(torch, y0)
                    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, dy, strategy=strat)
                    torch.cuda.synchronize()
                    outs.append(dy.cpu().numpy())
            for o in outs[1:]:
                assert np.array_equal(o, outs[0]), (m, strat)
            if nnz:
                assert oracle.scaled_error(outs[0], ref, 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, (m, strat)
        spmv_acc_amd.release_plans(drp)
    hiplib.spmv_acc_reset_tunables()


def test_row_digest_edge_cases(torch_dev, oracle, hiplib):
    """The row-block kernel's row extents from the plan's digest (1-byte lengths + per-block bases, tunable rowlen forced on):
    rows of exactly 254 / 255 / 256 / 257 non-zeros (255 is the last length a byte holds; longer rows flag their block, which
    then reads rowptr), such rows first / last in their block, empty rows and empty blocks, a row count that is not a multiple
    of the rows per block, every lane width, and a row shard that is not rebased (rowptr[0] > 0)."""
    torch = torch_dev
    rng = np.random.default_rng(77)
    m = 5003
    lens = rng.integers(0, 7, size=m)
    lens[0], lens[255], lens[256], lens[511] = 255, 254, 256, 257     # block edges of the 256-row blocks
    lens[1000:1300] = 0                                               # an empty block and more
    lens[2047], lens[2048] = 255, 255
    lens[m - 1] = 256
    rowptr, cols, vals = synth.csr_from_row_lengths(lens, 6000, rng)
    nnz = int(rowptr[-1])
    x, y0 = rng.standard_normal(6000), rng.standard_normal(m)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
    ref = oracle.host_spmv(0.5, -2.0, rowptr, cols, vals, x, y0)
    try:
        for vec in (0, 1, 2, 4, 8, 16, 32, 64):
            for target in (1900, 300):
                hiplib.spmv_acc_reset_tunables()
                for k, val in (("rowlen", 1), ("rowblock_guard", 0), ("rowblock_vec", vec), ("rowblock_target", target)):
                    assert hiplib.spmv_acc_set_tunable(k.encode(), val) == 0
                spmv_acc_amd.release_plans(drp)
                for _ in range(2):  # both walking directions
                    dy = dev(torch, y0)
                    spmv_acc_amd.csr_spmv(0.5, -2.0, m, 6000, nnz, drp, dci, dv, dx, dy, strategy="line_enhance")
                    torch.cuda.synchronize()
                    err = oracle.scaled_error(dy.cpu().numpy(), ref, 0.5, -2.0, rowptr, cols, vals, x, y0)
                    assert err <= SCALED_TOL, (vec, target, err)
        # not rebased shard
        r0, r1 = 257, 4100
        hiplib.spmv_acc_reset_tunables()
        assert hiplib.spmv_acc_set_tunable(b"rowlen", 1) == 0 and hiplib.spmv_acc_set_tunable(b"rowblock_guard", 0) == 0
        dy = dev(torch, y0[r0:r1])
        spmv_acc_amd.csr_spmv(0.5, -2.0, r1 - r0, 6000, int(rowptr[r1]), drp[r0: r1 + 1], dci, dv, dx, dy, strategy="line_enhance")
        torch.cuda.synchronize()
        assert np.max(np.abs(dy.cpu().numpy() - ref[r0:r1])) <= 1e-11
        spmv_acc_amd.release_plans(drp[r0: r1 + 1])
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans(drp)


def test_flat_segmented_scan_reduction(torch_dev, oracle, hiplib):
    """flat with its tiles reduced by the segmented scan (tunable flat_reduce = 1; what segment_sum_flat_sparse_spmv runs; reference:
    hip-flat/flat.cpp:59-76 with block_segment_sum, common/utils.h:75-94): rows that start on a lane's first / last product, rows that span
    lanes, waves and tiles, runs of empty rows at tile edges, one row over many tiles, hypersparse tails, a single non-zero, both cut-row
    forms, both walking directions, alpha / beta classes."""
    torch = torch_dev
    rng = np.random.default_rng(404)
    cases = []
    lens = rng.integers(0, 12, size=9001)
    lens[17], lens[400], lens[401] = 2048, 5000, 1          # exactly one tile; several tiles; a one-product row right behind
    lens[1000:1400] = 0                                      # a run of empty rows
    lens[3000:3008] = 8                                      # rows equal to a lane's chunk
    lens[5000] = 511
    cases.append(("mixed", lens, 7000))
    cases.append(("one_long_row", np.array([0, 0, 30000, 0, 1]), 40000))
    cases.append(("ones", np.ones(10000, dtype=np.int64), 10000))
    cases.append(("hypersparse", (rng.random(200000) < 0.01).astype(np.int64), 5000))
    cases.append(("single", np.array([0, 1, 0]), 3))
    cases.append(("sevens_and_nines", np.where(np.arange(6000) % 2 == 0, 7, 9), 6000))
    cases.append(("wave_rows", np.full(700, 64), 4096))
    try:
        for name, lens, n in cases:
            rowptr, cols, vals = synth.csr_from_row_lengths(lens, n, rng)
            m, nnz = len(lens), int(rowptr[-1])
            x, y0 = rng.standard_normal(n), rng.standard_normal(m)
            drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
            for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.75, 2.5)):
                ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
                for finish in (-1, 0, 1):
                    hiplib.spmv_acc_reset_tunables()
                    assert hiplib.spmv_acc_set_tunable(b"flat_reduce", 1) == 0 and hiplib.spmv_acc_set_tunable(b"flat_finish", finish) == 0
                    spmv_acc_amd.release_plans(drp)
                    for _ in range(2):
                        dy = dev(torch, y0)
                        spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, dy, strategy="flat")
                        torch.cuda.synchronize()
                        err = oracle.scaled_error(dy.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0)
                        assert err <= SCALED_TOL, (name, alpha, beta, finish, err)
            spmv_acc_amd.release_plans(drp)
    finally:
        hiplib.spmv_acc_reset_tunables()


def test_gather_hints_change_no_bit(torch_dev, oracle, hiplib):
    """Gather hints (k_hint.hip: the plan's column census marks the non-zeros whose x line lies outside the hot set, and the row-block-plus /
    flat / row-block kernels gather those non-temporally) steer a cache policy only: with hints forced on, y is bit-identical to the plain kernels' and
    matches the oracle -- on power-law columns with hub rows (the long-row slices of row-block-plus), for hot sets from one line to
    everything, for nnz that is no multiple of 8, and for the automatic mode (census + timed choice) on a matrix too small to want them."""
    torch = torch_dev
    rng = np.random.default_rng(909)
    m, n = 30011, 200000
    lens = np.minimum(rng.zipf(1.7, size=m), 9000)
    lens[5] = 25000                                   # a hub row: several dedicated long-row blocks
    rowptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lens, out=rowptr[1:])
    nnz = int(rowptr[-1])
    cols = np.minimum((rng.pareto(0.9, size=nnz) * 40).astype(np.int64), n - 1).astype(np.int32)   # a few hot columns, a long cold tail
    vals = rng.uniform(-1.0, 1.0, size=nnz)
    rowptr = rowptr.astype(np.int32)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
    ref = oracle.host_spmv(0.5, -2.0, rowptr, cols, vals, x, y0)
    try:
        # (the plans' other timed choices -- block size, cut-row form -- are pinned: they regroup sums, and a fresh plan may time them differently)
        for strat, base in (("adaptive_plus", {"plus_min_nnz": 1024}), ("adaptive_plus", {"plus_min_nnz": 1920}),
                            ("line_enhance", {"rowblock_guard": 0}), ("line_enhance", {"rowblock_guard": 0, "rowlen": 1, "rowblock_vec": 2}),
                            ("line", {"rowblock_guard": 0, "rowblock_vec": 16}),
                            ("flat", {"flat_npt": 8, "flat_early": 0, "flat_finish": 1}), ("flat", {"flat_npt": 8, "flat_early": 0, "flat_finish": 0})):
            outs = {}
            for tag, knobs in (("plain", {"gather_hint": 0}), ("one_line", {"gather_hint": 1, "hint_budget_kb": 1}),
                               ("l2", {"gather_hint": 1, "hint_budget_kb": 64}), ("all", {"gather_hint": 1, "hint_budget_kb": 1 << 20}),
                               ("auto", {"gather_hint": -1})):
                hiplib.spmv_acc_reset_tunables()
                for k, val in {**base, **knobs}.items():
                    assert hiplib.spmv_acc_set_tunable(k.encode(), val) == 0, k
                spmv_acc_amd.release_plans(drp)
                for _ in range(2):  # both walking directions
                    dy = dev(torch, y0)
                    spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, dy, strategy=strat)
                    torch.cuda.synchronize()
                    got = dy.cpu().numpy()
                    assert oracle.scaled_error(got, ref, 0.5, -2.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, (strat, tag)
                outs[tag] = got
            for tag, got in outs.items():
                assert np.array_equal(got, outs["plain"]), (strat, base, tag)
        # stale bits: colindex edited in place (same rowptr) under a live plan whose cold bits were derived from the old columns -- the plan is
        # kept (the guard watches rowptr only), the bits now describe other columns, and the result must still be the new matrix' product
        hiplib.spmv_acc_reset_tunables()
        assert hiplib.spmv_acc_set_tunable(b"gather_hint", 1) == 0 and hiplib.spmv_acc_set_tunable(b"hint_budget_kb", 64) == 0
        spmv_acc_amd.release_plans(drp)
        cols2 = np.sort(rng.integers(0, n, size=nnz)).astype(np.int32)  # any valid columns
        ref2 = oracle.host_spmv(0.5, -2.0, rowptr, cols2, vals, x, y0)
        for strat in ("adaptive_plus", "flat", "line_enhance"):
            dci.copy_(dev(torch, cols))
            dy = dev(torch, y0)
            spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, dy, strategy=strat)   # builds the plan and its bits from `cols`
            torch.cuda.synchronize()
            dci.copy_(dev(torch, cols2))                                                          # ... which now describe other columns
            dy = dev(torch, y0)
            spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, dy, strategy=strat)
            torch.cuda.synchronize()
            assert hiplib.spmv_acc_last_error() == 0
            assert oracle.scaled_error(dy.cpu().numpy(), ref2, 0.5, -2.0, rowptr, cols2, vals, x, y0) <= SCALED_TOL, strat
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans(drp)


def test_calls_before_a_plan_settles_are_bitwise_equal(torch_dev, oracle, hiplib):
    """Round 6 (review item 7): while a matrix's per-matrix timings are open its calls are served by the plan's RULE TWIN (what `deterministic = 1`
    computes) and the timings advance beside them against a scratch y; from the first settled call on the timed choices serve.  So y changes its last
    bits at most ONCE over the life of a plan, at a call the caller can see (`settled`), instead of whenever a timing phase finished (rounds 2-5).
    Checked for two strategies and both beta classes on a matrix big enough that settling takes several calls; `deterministic = -1` keeps the old
    behaviour and still matches the oracle."""
    torch = torch_dev
    m = n = 300_000
    drp, dci, dv = synth.structured_csr_torch(m, n, 7_500_000, 0x5E77, device="cuda")
    rowptr, cols, vals = (t.cpu().numpy() for t in (drp, dci, dv))
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    dx, dy0 = dev(torch, x), dev(torch, y0)

    def call(strategy, alpha, beta):
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y, strategy=strategy)
        torch.cuda.synchronize()
        assert hiplib.spmv_acc_last_error() == 0
        return y, spmv_acc_amd.query_plan(drp, m)["settled"]

    try:
        for strategy in ("adaptive", "flat"):
            for alpha, beta in ((0.5, -2.0), (1.25, 0.0)):
                ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
                # what the rule alone computes (a plan of its own: released before the run under test)
                spmv_acc_amd.release_plans(drp)
                assert hiplib.spmv_acc_set_tunable(b"deterministic", 1) == 0
                by_rule, _ = call(strategy, alpha, beta)
                assert hiplib.spmv_acc_set_tunable(b"deterministic", 0) == 0
                spmv_acc_amd.release_plans(drp)
                ys, settled = [], []
                for _ in range(60):
                    y, s = call(strategy, alpha, beta)
                    ys.append(y)
                    settled.append(s)
                    if len(settled) >= 4 and all(settled[-3:]):
                        break
                assert settled[-1] == 1, (strategy, beta, settled)
                first = settled.index(1)  # the call whose tuning pass closed the plan: itself still served by the twin
                assert first >= 1, (strategy, beta, settled)  # (several calls: what the test is about)
                for k in range(first + 1):
                    assert torch.equal(ys[k], by_rule), (strategy, beta, k, settled)
                for k in range(first + 1, len(ys)):
                    assert torch.equal(ys[k], ys[first + 1]), (strategy, beta, k, settled)
                    assert oracle.scaled_error(ys[k].cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL
                assert oracle.scaled_error(by_rule.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL
                assert hiplib.spmv_acc_cached_plans() >= 1
        # a settled kind of call stays with the timed choices while ANOTHER kind (the other beta class, another strategy) is still being timed
        spmv_acc_amd.release_plans(drp)
        ys = []
        for _ in range(60):
            y, s = call("adaptive", 0.5, -2.0)
            if s:
                break
        settled_y, _ = call("adaptive", 0.5, -2.0)
        for other in (("adaptive", 1.25, 0.0), ("line_enhance", 0.5, -2.0)):
            for _ in range(3):  # (unsettled calls of the other kind: served by the twin, their timings advancing)
                call(*other)
                again, _ = call("adaptive", 0.5, -2.0)
                assert torch.equal(again, settled_y), other
        # the earlier behaviour on request
        spmv_acc_amd.release_plans(drp)
        assert hiplib.spmv_acc_set_tunable(b"deterministic", -1) == 0
        for _ in range(6):
            y, _s = call("adaptive", 0.5, -2.0)
            ref = oracle.host_spmv(0.5, -2.0, rowptr, cols, vals, x, y0)
            assert oracle.scaled_error(y.cpu().numpy(), ref, 0.5, -2.0, rowptr, cols, vals, x, y0) <= SCALED_TOL
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans(drp)
