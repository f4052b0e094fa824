"""GPU suite (-m gpu): the out-of-place entry, per-thread library streams and stream captures, stale-plan attribution, per-beta-class choices,
shard handles and the pipelined step on a one-rank communicator, the tune cache across processes, the column-slab forms (copy and run lists),
LIGHT / BLOCK_ROW_ORDINARY, the first-call budget, the chunks entry.  Same tolerances as tests/test_gpu_parity.py (scaled error <= 1e-12 against the
CPU oracle; bit-exact where two library paths must agree).  (Until round 6: test_gpu_round3.py.)"""
import threading

import time

import numpy as np
import pytest

import spmv_acc_amd
from spmv_acc_amd import synth

pytestmark = pytest.mark.gpu

SCALED_TOL = 1e-12
ALL = spmv_acc_amd.STRATEGIES


@pytest.fixture(scope="module")
def torch_dev(hiplib):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def dev(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ---- out-of-place entry (spmv_acc_csr_spmv_oop) ------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,m,avg", [("powerlaw", 40000, 9), ("uniform", 60000, 5), ("uniform", 9000, 70), ("longrows", 3000, 40)])
def test_out_of_place_is_bitwise_the_in_place_result(torch_dev, oracle, hiplib, kind, m, avg):
    """y_out = alpha*A*x + beta*y_in through every strategy: bit-identical to the in-place entry run on a copy of y_in (same
    kernels, same sums), y_in untouched, and within tolerance of the oracle.  Flat is run in both of its cut-row forms (carries +
    fix-up kernel, rows finished in the tile) and adaptive-plus with long rows sliced over blocks -- the kernels that read the old y
    in a second kernel."""
    torch = torch_dev
    if kind == "longrows":
        rowptr, cols, vals = synth.random_csr(m, 50000, avg, seed=5, kind="uniform")
        # a few rows of tens of thousands of non-zeros: sliced by adaptive-plus, carried across many flat tiles
        rng = np.random.default_rng(3)
        lens = np.diff(rowptr).astype(np.int64)
        lens[[7, m // 2, m - 2]] = (30000, 9000, 20001)
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        nnz = int(rowptr[-1])
        cols = rng.integers(0, 50000, nnz).astype(np.int32)
        vals = rng.standard_normal(nnz)
        n = 50000
    else:
        rowptr, cols, vals = synth.random_csr(m, m, avg, seed=21, kind=kind)
        n = m
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(8)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    try:
        for alpha, beta in ((1.0, 1.0), (0.5, -2.0), (2.0, 0.0)):
            ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
            for strat in ALL:
                variants = [{}]
                if strat == "flat":
                    variants = [{"flat_finish": 0}, {"flat_finish": 1}, {"flat_reduce": 1}]
                for tun in variants:
                    for k, v in tun.items():
                        hiplib.spmv_acc_set_tunable(k.encode(), v)
                    try:
                        # (a settled plan: two calls on a plan whose per-matrix timings are still being finished -- tunable first_call_budget --
                        # may run different kernel families, i.e. sum in a different order)
                        spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy=strat, beta=beta)
                        inplace = dy0.clone()
                        spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, inplace, strategy=strat)
                        y_in = dy0.clone()
                        y_out = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
                        spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y_out, strategy=strat, y_in=y_in)
                        torch.cuda.synchronize()
                        tag = (strat, tun, alpha, beta)
                        assert torch.equal(y_in, dy0), (tag, "y_in was written")
                        assert torch.equal(y_out, inplace), (tag, "out-of-place differs from in-place")
                        got = y_out.cpu().numpy()
                        assert oracle.scaled_error(got, ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, tag
                    finally:
                        hiplib.spmv_acc_reset_tunables()
    finally:
        spmv_acc_amd.release_plans(drp)


def test_out_of_place_edge_cases(torch_dev, oracle, hiplib):
    """y_in == y_out is the in-place call; partially overlapping vectors are refused and nothing is written; a matrix without
    non-zeros scales y_in into y_out; beta == 0 never reads y_in (NaNs there do not reach y_out)."""
    torch = torch_dev
    m = n = 5000
    rowptr, cols, vals = synth.random_csr(m, n, 6, seed=2, kind="uniform")
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(4)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
    try:
        ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
        y = dev(torch, y0)
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive", y_in=y)
        torch.cuda.synchronize()
        assert oracle.scaled_error(y.cpu().numpy(), ref, 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL
        big = torch.zeros(m + 8, dtype=torch.float64, device="cuda")
        big[:m].copy_(dev(torch, y0))
        before = big.clone()
        with pytest.raises(spmv_acc_amd.SpmvAccError, match="overlap"):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, big[8:], strategy="flat", y_in=big[:m])
        torch.cuda.synchronize()
        assert torch.equal(big, before)
        hiplib.spmv_acc_clear_error()
        # beta == 0: y_in is not read
        poison = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
        ref0 = oracle.host_spmv(1.5, 0.0, rowptr, cols, vals, x, y0)
        for strat in ("adaptive", "flat", "line_enhance", "adaptive_plus", "vector_row", "wf_row"):
            out = torch.empty(m, dtype=torch.float64, device="cuda")
            spmv_acc_amd.csr_spmv(1.5, 0.0, m, n, nnz, drp, dci, dv, dx, out, strategy=strat, y_in=poison)
            torch.cuda.synchronize()
            assert oracle.scaled_error(out.cpu().numpy(), ref0, 1.5, 0.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, strat
    finally:
        spmv_acc_amd.release_plans(drp)
    # no non-zeros at all: y_out = beta * y_in
    erp = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
    eci = torch.zeros(1, dtype=torch.int32, device="cuda")
    ev = torch.zeros(1, dtype=torch.float64, device="cuda")
    y_in = dev(torch, y0)
    y_out = torch.zeros(m, dtype=torch.float64, device="cuda")
    spmv_acc_amd.csr_spmv(3.0, -0.5, m, n, 0, erp, eci, ev, dx, y_out, strategy="adaptive", y_in=y_in)
    torch.cuda.synchronize()
    assert np.array_equal(y_out.cpu().numpy(), -0.5 * y0) and np.array_equal(y_in.cpu().numpy(), y0)
    spmv_acc_amd.release_plans(erp)


# ---- per-thread library stream -----------------------------------------------------------------------------------------------
def _two_concurrent_streams(torch):
    """Two torch streams whose kernels really run side by side.  HIP multiplexes its streams onto a few hardware queues, round robin in creation
    order: two streams that land on the SAME queue serialise, and which two do depends on how many streams earlier tests created (this test failed in
    some suite orders and passed alone, on round 5's tree as on this one).  So: make a handful, and keep the first pair where a tiny kernel on one
    finishes while the other sleeps."""
    pool = [torch.cuda.Stream() for _ in range(8)]
    probe = torch.zeros(1, device="cuda")
    for a in range(len(pool)):
        for b in range(a + 1, len(pool)):
            torch.cuda.synchronize()
            with torch.cuda.stream(pool[a]):
                torch.cuda._sleep(int(1e8))  # ~40 ms
            with torch.cuda.stream(pool[b]):
                probe.add_(1.0)
            pool[b].synchronize()
            concurrent = not pool[a].query()
            torch.cuda.synchronize()
            if concurrent:
                return [pool[a], pool[b]]
    pytest.skip("no two streams of this process run concurrently (hardware queues exhausted)")


def test_two_host_threads_two_streams(torch_dev, oracle, hiplib):
    """The library stream belongs to the calling host thread.  Two threads, each with its own non-NULL stream and its own matrix,
    call concurrently (ctypes releases the GIL): every launch lands on its thread's stream -- checked by holding ONE of the streams
    back with a long sleep kernel: the other thread's results are complete while the held stream's y is still untouched -- and every
    result is right.  A third thread that never set a stream sees NULL.

    Runs in a FRESH process (the test re-invokes itself through pytest): late in a long process -- after the suite's one-rank RCCL tests and a few dozen
    streams -- the runtime makes this process' library launches on one stream wait for the other stream's sleep although plain torch kernels on the same
    two streams still run side by side; seen in some suite orders only, on round 5's tree as on this one (the control experiment below tells the two
    apart).  What the test is about -- which stream a thread's launches go to -- does not depend on the process's history."""
    import os
    import subprocess
    import sys

    if os.environ.get("SPMV_ACC_TEST_CHILD") != "two_streams":
        r = subprocess.run([sys.executable, "-m", "pytest", __file__ + "::test_two_host_threads_two_streams", "-x", "-q", "-p", "no:cacheprovider"],
                           env=dict(os.environ, SPMV_ACC_TEST_CHILD="two_streams"), capture_output=True, text=True, timeout=600,
                           cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-1500:])
        return
    torch = torch_dev
    torch.cuda.empty_cache()
    mats = []
    for seed, m, avg in ((1, 40000, 7), (2, 52000, 12)):
        rowptr, cols, vals = synth.random_csr(m, m, avg, seed=seed, kind="powerlaw")
        rng = np.random.default_rng(seed)
        x, y0 = rng.standard_normal(m), rng.standard_normal(m)
        mats.append(dict(rowptr=rowptr, cols=cols, vals=vals, x=x, y0=y0, m=m, nnz=int(rowptr[-1]),
                         d=[dev(torch, a) for a in (rowptr, cols, vals, x)], dy0=dev(torch, y0)))
    streams = _two_concurrent_streams(torch)
    seen_streams, results, errors = [None, None], [None, None], []
    go = threading.Barrier(2)
    held = threading.Event()
    slow_calls = []

    def worker(i):
        try:
            A = mats[i]
            drp, dci, dv, dx = A["d"]
            with torch.cuda.stream(streams[i]):
                # plans first, every per-matrix timing settled (calls that still tune synchronise), one strategy of each kernel family
                for strat in ("adaptive", "flat", "adaptive_plus"):
                    y = A["dy0"].clone()
                    spmv_acc_amd.csr_spmv(1.0, 1.0, A["m"], A["m"], A["nnz"], drp, dci, dv, dx, y, strategy=strat)
                    spmv_acc_amd.prepare(A["m"], A["m"], A["nnz"], drp, dci, dv, dx, strategy=strat)
                # every buffer of the contended section exists before it starts: an allocation there may free cached blocks,
                # and hipFree waits for the whole device -- including the other thread's sleeping stream
                outs = [torch.empty(A["m"], dtype=torch.float64, device="cuda") for _ in range(30)]
                streams[i].synchronize()
                go.wait()
                if i == 0:
                    torch.cuda._sleep(int(2e9))  # ~1 s on this stream only
                    held.set()
                else:
                    held.wait()
                t_section = time.perf_counter()
                for it in range(30):
                    y = outs[it]
                    t_call = time.perf_counter()
                    y.copy_(A["dy0"])  # (device-side copy on this thread's stream)
                    t_copy = time.perf_counter()
                    spmv_acc_amd.csr_spmv(1.0, 1.0, A["m"], A["m"], A["nnz"], drp, dci, dv, dx, y,
                                          strategy=("adaptive", "flat", "adaptive_plus")[it % 3])
                    if time.perf_counter() - t_call > 0.05:  # a host call that waited (for the other stream's sleep?): say which
                        slow_calls.append((i, it, ("adaptive", "flat", "adaptive_plus")[it % 3], round(t_copy - t_call, 3), round(time.perf_counter() - t_copy, 3),
                                           hiplib.spmv_acc_last_prepare_us()))
                seen_streams[i] = hiplib.spmv_acc_get_stream()
                if i == 1:
                    streams[1].synchronize()  # must not wait for stream 0's sleep
                    results[1] = [o.cpu().numpy() for o in outs]
                    # stream 0 is still asleep: had thread 0's launches gone to this thread's stream they would be done now
                    results[0] = "pending" if not streams[0].query() else f"stream 0 already idle ({time.perf_counter() - t_section:.3f} s after thread 1 started its 30 calls)"
                else:
                    streams[0].synchronize()
                    results[0] = [o.cpu().numpy() for o in outs]
        except Exception as ex:  # noqa: BLE001
            errors.append((i, repr(ex)))

    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    ts[1].join()
    pending_seen = results[0]
    ts[0].join()
    try:
        assert not errors, errors
        if pending_seen != "pending":
            # CONTROL before blaming the library: the same shape of work without it -- a sleep on stream 0, thirty plain torch kernels on stream 1.
            # If those cannot finish while stream 0 sleeps either, this process' streams are being serialised by the runtime (seen after earlier
            # tests of the suite have used RCCL, on round 5's tree as on this one; the pair was concurrent when _two_concurrent_streams probed it)
            # and the "pending" evidence cannot be had here: the routing and the results below are still checked.
            torch.cuda.synchronize()
            scratch = torch.zeros(50_000, dtype=torch.float64, device="cuda")
            with torch.cuda.stream(streams[0]):
                torch.cuda._sleep(int(5e8))
            with torch.cuda.stream(streams[1]):
                for _ in range(30):
                    scratch.add_(1.0)
            streams[1].synchronize()
            serialised_by_the_runtime = streams[0].query()
            torch.cuda.synchronize()
            assert serialised_by_the_runtime, ("thread 1's library calls waited for stream 0 although plain kernels on the same two streams run side by side",
                                               pending_seen, slow_calls)
            print("test_two_host_threads_two_streams: the runtime serialises the two streams in this process (control experiment); "
                  "the held-stream evidence is void, routing and results are checked", flush=True)
        assert seen_streams[0] == streams[0].cuda_stream and seen_streams[1] == streams[1].cuda_stream
        for i in range(2):
            A = mats[i]
            ref = oracle.host_spmv(1.0, 1.0, A["rowptr"], A["cols"], A["vals"], A["x"], A["y0"])
            for got in results[i]:
                assert oracle.scaled_error(got, ref, 1.0, 1.0, A["rowptr"], A["cols"], A["vals"], A["x"], A["y0"]) <= SCALED_TOL, i
        other = []
        t = threading.Thread(target=lambda: other.append(hiplib.spmv_acc_get_stream()))
        t.start()
        t.join()
        assert other == [None]
    finally:
        for A in mats:
            spmv_acc_amd.release_plans(A["d"][0])


# ---- plan work inside a stream capture ----------------------------------------------------------------------
def test_capture_of_an_unprepared_strategy_is_refused_not_broken(torch_dev, oracle, hiplib):
    """A matrix prepared with strategy A and then captured with strategy B: B's structural plan work (break points, row-block
    analysis) would allocate and synchronise inside the capture.  The call enqueues nothing and reports
    SPMV_ACC_ERR_BAD_ARGUMENT naming the cause; the capture itself stays valid (it ends cleanly and replays); after one call of B
    outside a capture B captures and replays bit-exactly.  Merely missing TIMED choices do not refuse the call: a matrix prepared
    at beta = 1 captures at beta = 0."""
    torch = torch_dev
    m = n = 30000
    # (evenly filled rows: line_enhance keeps its fixed row blocks, so neither flat's break points nor the row-block analysis exist yet)
    rowptr, cols, vals = synth.random_csr(m, n, 9, seed=77, kind="uniform")
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(5)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    side = torch.cuda.Stream()
    try:
        with torch.cuda.stream(side):
            spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy="line_enhance")
        for other in ("flat", "adaptive_plus", "adaptive"):
            static_y = dy0.clone()
            marker = torch.zeros(4, device="cuda")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                marker.add_(1.0)  # something the capture does record
                with pytest.raises(spmv_acc_amd.SpmvAccError, match="capture"):
                    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, static_y, strategy=other)
            hiplib.spmv_acc_clear_error()
            g.replay()
            torch.cuda.synchronize()
            assert float(marker[0].item()) == 1.0, "the capture was invalidated"
            assert torch.equal(static_y, dy0), "a refused call wrote y"
            # once outside a capture, then it captures
            with torch.cuda.stream(side):
                eager = dy0.clone()
                spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, eager, strategy=other)
            side.synchronize()
            g2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, stream=side):
                spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, static_y, strategy=other)
            static_y.copy_(dy0)
            g2.replay()
            torch.cuda.synchronize()
            assert torch.equal(static_y, eager), other
            # the other beta class was never timed: captured all the same (choices fall back to the timed class)
            g3 = torch.cuda.CUDAGraph()
            z = torch.zeros(m, dtype=torch.float64, device="cuda")
            with torch.cuda.graph(g3, stream=side):
                spmv_acc_amd.csr_spmv(2.0, 0.0, m, n, nnz, drp, dci, dv, dx, z, strategy=other)
            g3.replay()
            torch.cuda.synchronize()
            ref0 = oracle.host_spmv(2.0, 0.0, rowptr, cols, vals, x, y0)
            assert oracle.scaled_error(z.cpu().numpy(), ref0, 2.0, 0.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, other
    finally:
        hiplib.spmv_acc_set_stream(None)
        spmv_acc_amd.release_plans(drp)


def test_one_matrix_on_two_streams_is_ordered(torch_dev, oracle, hiplib):
    """A plan owns scratch its kernels write (flat's carries, row-block-plus partials, the slab passes' partial sums, LIGHT's counter).  Two
    SpMVs of ONE matrix enqueued on two different streams would share it while both run; the engine orders them (the call on the other
    stream waits for an event behind the plan's previous launches).  Many alternating launches on two streams, different x and y per
    stream, no synchronisation in between: every result is right."""
    torch = torch_dev
    m, n = 30000, 30000
    rowptr, cols, vals = synth.random_csr(m, n, 40, seed=91, kind="uniform")
    lens = np.diff(rowptr).astype(np.int64)
    lens[[5, m // 2]] = (9000, 4100)  # rows cut across many tiles / sliced over row blocks
    rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(17)
    cols = np.sort(rng.integers(0, n, nnz).astype(np.int32).reshape(-1))  # (sorted globally is sorted per row too: the run lists apply)
    cols = np.concatenate([np.sort(cols[rowptr[i]:rowptr[i + 1]]) for i in range(m)]).astype(np.int32)
    vals = rng.standard_normal(nnz)
    xs = [rng.standard_normal(n) for _ in range(2)]
    y0 = rng.standard_normal(m)
    drp, dci, dv = (dev(torch, a) for a in (rowptr, cols, vals))
    dxs = [dev(torch, x) for x in xs]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    refs = [oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0) for x in xs]
    try:
        for strat, knobs in (("flat", {"flat_finish": 0}), ("adaptive_plus", {}), ("light", {}), ("line_enhance", {"slab_segments": 4})):
            hiplib.spmv_acc_reset_tunables()
            for k, val in knobs.items():
                hiplib.spmv_acc_set_tunable(k.encode(), val)
            warm = dev(torch, y0)
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dxs[0], warm, strategy=strat)  # plan + timings on the NULL stream
            torch.cuda.synchronize()
            ys = [[dev(torch, y0) for _ in range(12)] for _ in range(2)]
            sid = spmv_acc_amd.strategy_id(strat)
            for k in range(12):
                for s in (0, 1):
                    hiplib.spmv_acc_set_stream(streams[s].cuda_stream)
                    hiplib.spmv_acc_csr_spmv_strategy(sid, 0, 1.0, 1.0, m, n, nnz, None, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(),
                                                      dxs[s].data_ptr(), ys[s][k].data_ptr())
            torch.cuda.synchronize()
            assert hiplib.spmv_acc_last_error() == 0, hiplib.spmv_acc_last_error_string()
            for s in (0, 1):
                for k in range(12):
                    err = oracle.scaled_error(ys[s][k].cpu().numpy(), refs[s], 1.0, 1.0, rowptr, cols, vals, xs[s], y0)
                    assert err <= SCALED_TOL, (strat, s, k, err)
            spmv_acc_amd.release_plans(drp)
    finally:
        hiplib.spmv_acc_set_stream(None)
        hiplib.spmv_acc_reset_tunables()
        hiplib.spmv_acc_clear_error()
        spmv_acc_amd.release_plans()


# ---- stale-plan attribution ----------------------------------------------------------------------------------
def test_stale_plan_is_reported_to_the_thread_that_used_it(torch_dev, hiplib):
    """spmv_acc_last_error() asks the plan the CALLING thread used last and nothing else: a plan made stale by thread A is not
    reported to thread B working on another matrix; thread A gets it (after synchronising), and spmv_acc_check_plans() finds it
    from any thread."""
    torch = torch_dev
    m = n = 20000

    def matrix(seed):
        rp, ci, v = synth.random_csr(m, n, 8, seed=seed, kind="uniform")
        return rp, ci, v

    rpA, ciA, vA = matrix(1)
    lens = np.diff(rpA)[::-1].copy()  # same nnz, other structure
    rpA2 = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    rpB, ciB, vB = matrix(2)
    dA = [dev(torch, a) for a in (rpA, ciA, vA)]
    dB = [dev(torch, a) for a in (rpB, ciB, vB)]
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    out = {}

    def thread_a():
        y = torch.zeros(m, dtype=torch.float64, device="cuda")
        spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, int(rpA[-1]), *dA, x, y, strategy="flat")
        torch.cuda.synchronize()
        dA[0].copy_(dev(torch, rpA2))  # structure rewritten in place, no release
        torch.cuda.synchronize()
        hiplib.spmv_acc_csr_spmv_strategy(spmv_acc_amd.strategy_id("flat"), 0, 1.0, 0.0, m, n, int(rpA[-1]), None,
                                          dA[0].data_ptr(), dA[1].data_ptr(), dA[2].data_ptr(), x.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        out["a_ready"] = True

    def thread_b():
        y = torch.zeros(m, dtype=torch.float64, device="cuda")
        spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, int(rpB[-1]), *dB, x, y, strategy="flat")
        torch.cuda.synchronize()
        out["b_err"] = hiplib.spmv_acc_last_error()
        out["b_plans"] = hiplib.spmv_acc_cached_plans()

    try:
        hiplib.spmv_acc_clear_error()
        ta = threading.Thread(target=thread_a)
        ta.start()
        ta.join()
        tb = threading.Thread(target=thread_b)
        tb.start()
        tb.join()
        assert out["b_err"] == 0 and out["b_plans"] == 2, out  # thread B: its own plan is fine, A's stale plan is not its business
        assert hiplib.spmv_acc_last_error() == 0  # nor the main thread's (it has used no plan)
        assert hiplib.spmv_acc_check_plans() == 1  # the explicit all-plans check finds and drops it
        assert hiplib.spmv_acc_last_error() == 2 and b"changed" in hiplib.spmv_acc_last_error_string()
        hiplib.spmv_acc_clear_error()
        assert hiplib.spmv_acc_cached_plans() == 1
    finally:
        hiplib.spmv_acc_clear_error()
        spmv_acc_amd.release_plans()


@pytest.mark.parametrize("strat", ["flat", "line_enhance", "adaptive_plus"])
def test_guard_full_notices_an_edit_between_the_samples(torch_dev, oracle, hiplib, strat):
    """The guard every kernel carries compares 64 strided rowptr samples: moving one non-zero from a row to its neighbour at an
    index that is NOT a sample goes unnoticed (documented window).  Tunable guard_full = 1 re-reads all of rowptr on every call:
    the same edit raises the stale-plan error on the call that met it, the plan is rebuilt, and the next call is right."""
    torch = torch_dev
    m, n = 50000, 50000
    rowptr, cols, vals = synth.random_csr(m, n, 9, seed=77, kind="uniform")
    nnz = int(rowptr[-1])
    samples = {int(k * m // 63) for k in range(64)}
    i = next(r for r in range(m // 3, m) if r not in samples and rowptr[r] - rowptr[r - 1] >= 2)
    edited = rowptr.copy()
    edited[i] -= 1  # row i takes over the last non-zero of row i-1: same nnz, same 64 samples
    assert all(edited[int(k * m // 63)] == rowptr[int(k * m // 63)] for k in range(64))
    x = np.random.default_rng(2).standard_normal(n)
    y0 = np.zeros(m)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
    sid = spmv_acc_amd.strategy_id(strat)

    def call(y):
        hiplib.spmv_acc_csr_spmv_strategy(sid, 0, 1.0, 0.0, m, n, nnz, None, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(),
                                          dx.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        return hiplib.spmv_acc_last_error()

    try:
        for full in (0, 1):
            drp.copy_(dev(torch, rowptr))
            torch.cuda.synchronize()
            spmv_acc_amd.release_plans()
            hiplib.spmv_acc_clear_error()
            hiplib.spmv_acc_set_tunable(b"guard_full", full)
            y = torch.zeros(m, dtype=torch.float64, device="cuda")
            assert call(y) == 0
            ref = oracle.host_spmv(1.0, 0.0, rowptr, cols, vals, x, y0)
            assert oracle.scaled_error(y.cpu().numpy(), ref, 1.0, 0.0, rowptr, cols, vals, x, y0) <= SCALED_TOL
            assert call(y) == 0  # an unchanged matrix passes the full check too
            drp.copy_(dev(torch, edited))  # in place, not announced
            torch.cuda.synchronize()
            err = call(y)
            if full == 0:
                assert err == 0  # the window: nothing noticed (y may or may not be right, depending on what the plan holds)
                continue
            assert err == 2 and b"changed" in hiplib.spmv_acc_last_error_string()
            hiplib.spmv_acc_clear_error()
            assert call(y) == 0  # fresh plan for the edited structure
            ref = oracle.host_spmv(1.0, 0.0, edited, cols, vals, x, y0)
            assert oracle.scaled_error(y.cpu().numpy(), ref, 1.0, 0.0, edited, cols, vals, x, y0) <= SCALED_TOL
    finally:
        hiplib.spmv_acc_set_tunable(b"guard_full", 0)
        hiplib.spmv_acc_clear_error()
        spmv_acc_amd.release_plans()


def test_guard_full_replays_from_a_graph(torch_dev, oracle, hiplib):
    """The per-call digest + verdict pair is two ordinary launches: a captured SpMV with guard_full = 1 replays (the accumulator is
    left at zero by the verdict kernel), and an in-place edit between replays is noticed by the replay that meets it."""
    torch = torch_dev
    m, n = 30000, 30000
    rowptr, cols, vals = synth.random_csr(m, n, 7, seed=78, kind="uniform")
    nnz = int(rowptr[-1])
    x = np.random.default_rng(3).standard_normal(n)
    y0 = np.zeros(m)
    drp, dci, dv, dx = (dev(torch, a) for a in (rowptr, cols, vals, x))
    y = torch.zeros(m, dtype=torch.float64, device="cuda")
    samples = {int(k * m // 63) for k in range(64)}
    i = next(r for r in range(m // 2, m) if r not in samples and rowptr[r] - rowptr[r - 1] >= 2)
    edited = rowptr.copy()
    edited[i] -= 1
    stream = torch.cuda.Stream()
    try:
        hiplib.spmv_acc_set_tunable(b"guard_full", 1)
        hiplib.spmv_acc_clear_error()
        with torch.cuda.stream(stream):
            hiplib.spmv_acc_set_stream(stream.cuda_stream)
            spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, nnz, drp, dci, dv, dx, y, strategy="line_enhance")  # plan + digest outside the capture
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                hiplib.spmv_acc_csr_spmv_strategy(spmv_acc_amd.strategy_id("line_enhance"), 0, 1.0, 0.0, m, n, nnz, None, drp.data_ptr(),
                                                  dci.data_ptr(), dv.data_ptr(), dx.data_ptr(), y.data_ptr())
            assert hiplib.spmv_acc_last_error() == 0, hiplib.spmv_acc_last_error_string()
        ref = oracle.host_spmv(1.0, 0.0, rowptr, cols, vals, x, y0)
        for _ in range(3):
            y.zero_()
            g.replay()
            torch.cuda.synchronize()
            assert hiplib.spmv_acc_last_error() == 0
            assert oracle.scaled_error(y.cpu().numpy(), ref, 1.0, 0.0, rowptr, cols, vals, x, y0) <= SCALED_TOL
        drp.copy_(dev(torch, edited))
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        assert hiplib.spmv_acc_last_error() == 2 and b"changed" in hiplib.spmv_acc_last_error_string()
    finally:
        hiplib.spmv_acc_set_stream(None)
        hiplib.spmv_acc_set_tunable(b"guard_full", 0)
        hiplib.spmv_acc_clear_error()
        spmv_acc_amd.release_plans()


def test_adaptive_family_is_kept_per_beta_class(torch_dev, hiplib):
    """adaptive times the kernel families in the caller's beta class and keeps one choice per class (the ranking flips where rows
    hold one or two non-zeros); spmv_acc_prepare (beta = 1) leaves the beta == 0 class untimed until a beta == 0 call arrives."""
    torch = torch_dev
    m = n = 200000
    rowptr, cols, vals = synth.random_csr(m, n, 2, seed=9, kind="uniform")
    nnz = int(rowptr[-1])
    drp, dci, dv = (dev(torch, a) for a in (rowptr, cols, vals))
    x = torch.ones(n, dtype=torch.float64, device="cuda")
    try:
        spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, x, strategy="adaptive")
        info = spmv_acc_amd.query_plan(drp, m)
        assert info["adaptive_family"] in (0, 1, 2)
        assert hiplib.spmv_acc_query_plan_beta0(drp.data_ptr(), m) == -1
        y = torch.zeros(m, dtype=torch.float64, device="cuda")
        spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, nnz, drp, dci, dv, x, y, strategy="adaptive")
        torch.cuda.synchronize()
        assert hiplib.spmv_acc_query_plan_beta0(drp.data_ptr(), m) in (0, 1, 2)
        assert hiplib.spmv_acc_query_plan_beta0(torch.zeros(4, dtype=torch.int32, device="cuda").data_ptr(), 3) == -2
    finally:
        spmv_acc_amd.release_plans(drp)


# ---- row-sharded step: handle API of the C boundary, pipelined step ------------------------------------------------------------
def test_shard_handle_api_on_a_one_rank_communicator(torch_dev, oracle, hiplib):
    """spmv_acc_shard_create / _step / _destroy with a communicator from spmv_acc_rccl_comm_init_all (ncclCommInitAll bound at run
    time, as spmv-cli --gpus N uses it): the step computes this rank's rows straight into the gathered vector -- in place
    (dy_in NULL) and out of place (old slice elsewhere, untouched) -- and exchanges in place: one allgather (pipeline 1) or the
    chunked point-to-point form (pipeline 4: chunks as un-rebased row sub-ranges, their kernels alternating over two streams of the shard's own,
    an exchange stream, events); spmv_acc_shard_prepare settles every chunk's plan first.  One rank is all this box allows; every
    RCCL call that a one-rank communicator makes is made."""
    import ctypes

    torch = torch_dev
    torch.zeros(1, device="cuda")
    comm = ctypes.c_void_p()
    assert hiplib.spmv_acc_rccl_comm_init_all(ctypes.byref(comm), 1, None) == 0, hiplib.spmv_acc_last_error_string()
    side = torch.cuda.Stream()
    try:
        m, n, pad = 70_001, 64_000, 70_016
        rowptr, cols, vals = synth.random_csr(m, n, 9, seed=15, kind="powerlaw")
        nnz = int(rowptr[-1])
        rng = np.random.default_rng(16)
        x, y0 = rng.standard_normal(n), rng.standard_normal(m)
        drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
        hiplib.spmv_acc_set_stream(side.cuda_stream)
        for pipeline in (1, 4):
            for strat, (alpha, beta) in (("adaptive", (1.0, 1.0)), ("flat", (0.5, -2.0)), ("line_enhance", (2.0, 0.0))):
                shard = ctypes.c_void_p()
                rc = hiplib.spmv_acc_shard_create(ctypes.byref(shard), comm, spmv_acc_amd.strategy_id(strat), m, pad, n, nnz,
                                                  drp.data_ptr(), dci.data_ptr(), dv.data_ptr(), pipeline)
                assert rc == 0, hiplib.spmv_acc_last_error_string()
                assert hiplib.spmv_acc_shard_pipeline(shard) == pipeline
                # round 4: every chunk's plan built and tuned up front, outside any collective -- the steps below find nothing left to prepare
                assert hiplib.spmv_acc_shard_prepare(shard, beta, dx.data_ptr()) == 0, hiplib.spmv_acc_last_error_string()
                assert hiplib.spmv_acc_shard_prepare(shard, beta, None) == 2 and hiplib.spmv_acc_shard_prepare(None, beta, dx.data_ptr()) == 2
                hiplib.spmv_acc_clear_error()
                ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
                # out of place: the old slice stays where it is
                y_full = torch.zeros(pad, dtype=torch.float64, device="cuda")
                torch.cuda.synchronize()
                assert hiplib.spmv_acc_shard_step(shard, alpha, beta, dx.data_ptr(), dy0.data_ptr(), y_full.data_ptr()) == 0, \
                    hiplib.spmv_acc_last_error_string()
                assert hiplib.spmv_acc_last_prepare_us() == 0.0, (pipeline, strat, "a prepared shard's step did plan work")
                assert hiplib.spmv_acc_get_stream() == side.cuda_stream  # (the chunks ran on the shard's own streams; the caller's is back)
                side.synchronize()
                got = y_full.cpu().numpy()
                assert oracle.scaled_error(got[:m], ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (pipeline, strat)
                assert np.all(got[m:] == 0.0) and torch.equal(dy0, dev(torch, y0))
                # in place: the slice of the gathered vector is the old y
                y_full.zero_()
                y_full[:m].copy_(dy0)
                torch.cuda.synchronize()
                assert hiplib.spmv_acc_shard_step(shard, alpha, beta, dx.data_ptr(), None, y_full.data_ptr()) == 0
                side.synchronize()
                got = y_full.cpu().numpy()
                assert oracle.scaled_error(got[:m], ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (pipeline, strat, "in place")
                assert hiplib.spmv_acc_shard_destroy(shard) == 0
        bad = ctypes.c_void_p()
        assert hiplib.spmv_acc_shard_create(ctypes.byref(bad), None, 1, m, pad, n, nnz, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(), 1) == 2
        assert hiplib.spmv_acc_shard_create(ctypes.byref(bad), comm, 1, m, m - 1, n, nnz, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(), 1) == 2
        hiplib.spmv_acc_clear_error()
    finally:
        hiplib.spmv_acc_set_stream(None)
        spmv_acc_amd.release_plans()
        hiplib.spmv_acc_rccl_comm_destroy(comm)


def test_row_sharded_pipelined_step_on_one_rank_rccl(torch_dev, oracle):
    """RowShardedSpmv with pipeline = 3 over a one-rank RCCL process group: three chunk kernels on the engine's compute stream,
    each chunk's exchange issued behind ITS event; the compute stream is held back by a sleep, so an exchange (or the final wait)
    that did not order itself behind the kernels would hand back the old vector.  No device copy anywhere: the step writes
    straight into the gathered vector and the dependent iteration x_{k+1} = y_k reads the vector the step just completed."""
    import os
    import socket

    torch = torch_dev
    import torch.distributed as dist

    from spmv_acc_amd.dist import RowShardedSpmv

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        m = n = 150_000
        rowptr, cols, vals = synth.random_csr(m, n, 6, seed=19)
        nnz = int(rowptr[-1])
        rng = np.random.default_rng(4)
        x0, y0 = rng.standard_normal(n), rng.standard_normal(m)
        drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x0, y0))
        eng = RowShardedSpmv(0, 1, np.array([0, m], dtype=np.int64), drp, dci, dv, n, torch.device("cuda", 0), strategy="adaptive",
                             always_collective=True, exchange="allgather", pipeline=3)
        assert len(eng.chunk_bounds(3)) == 3 and eng.chunk_bounds(3)[-1][1] == m
        eng.set_y(dy0)
        eng.step(0.25, 0.5, dx, overlap=False)  # builds the three chunk plans
        torch.cuda.synchronize()
        x = eng.y_full[:n]  # dependent iteration: the next x IS the vector the step completed (no copy)
        with torch.cuda.stream(eng.compute_stream):
            torch.cuda._sleep(200_000_000)
        eng.step(0.25, 0.5, x, overlap=False)
        assert not eng.spmv_done.query() or True  # (the sleep may already be over on a fast box; the result check is what counts)
        got = eng.gathered().cpu().numpy()
        y1 = oracle.host_spmv(0.25, 0.5, rowptr, cols, vals, x0, y0)
        y2 = oracle.host_spmv(0.25, 0.5, rowptr, cols, vals, y1, y1)
        assert oracle.scaled_error(got, y2, 0.25, 0.5, rowptr, cols, vals, y1, y1) <= 1e-11
        timings = eng.tune_pipeline(0.25, 0.5, dx, candidates=(1, 2, 4), warm=1, iters=2)
        assert set(timings) == {1, 2, 4} and eng.pipeline in timings
        spmv_acc_amd.release_plans()
    finally:
        dist.destroy_process_group()
        spmv_acc_amd.load_library().spmv_acc_set_stream(None)


# ---- persistent choices and the deterministic switch -------------------------------------------------------------------------
_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
import spmv_acc_amd
from spmv_acc_amd import synth
rp, ci, v = synth.structured_csr_torch(600_000, 600_000, 4_200_000, 0xC7, device="cuda")
m = n = 600_000
nnz = int(rp[-1].item())
gen = torch.Generator(device="cuda"); gen.manual_seed(5)
x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
out = {{}}
# first every strategy's plan work (first call + the refinements its second call may still make), THEN the results: the strategies share one
# plan, a later strategy's timings (flat's tile size, say) are choices an earlier strategy (adaptive running flat) also uses from then on, and
# the second process adopts the FINAL record -- results taken in between would be compared with other choices than they were computed with
for strat in {strategies!r}:
    out[strat + "__prepare_ms"] = np.array(spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat))
    for _ in range(2):
        y = torch.ones(m, dtype=torch.float64, device="cuda")
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
torch.cuda.synchronize()
for strat in {strategies!r}:
    y = torch.ones(m, dtype=torch.float64, device="cuda")
    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    torch.cuda.synchronize()
    out[strat] = y.cpu().numpy()
    out[strat + "__plan"] = np.array(list((spmv_acc_amd.query_plan(rp, m) or {{}}).values()))
np.savez({out!r}, **out)
"""


def _run_child(tmp_path, tag, strategies, env_extra):
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = str(tmp_path / f"{tag}.npz")
    env = dict(os.environ, SPMV_ACC_TUNE_LOG="1", **env_extra)
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=root, strategies=strategies, out=out)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return np.load(out), r.stderr


def test_tune_cache_is_adopted_by_the_next_process(torch_dev, tmp_path):
    """SPMV_ACC_TUNE_CACHE=<file>: the first process times its choices on the matrix and appends them; a second process meeting
    the same matrix on the same device adopts them -- no timing line in its tune log, a cheaper first call, the same plan, and
    therefore bitwise the same y."""
    strategies = ("adaptive", "flat", "adaptive_plus")
    cache = str(tmp_path / "tune.txt")
    first, log1 = _run_child(tmp_path, "p1", strategies, {"SPMV_ACC_TUNE_CACHE": cache})
    assert "stream policy" in log1 and "adopted" not in log1
    lines = open(cache).read().splitlines()
    assert lines and all(ln.startswith("spmvacc6 ") and len(ln.split()) == 28 for ln in lines)
    # a damaged file costs at most the damaged lines: one cut short by a killed writer, one from another version, one of noise
    with open(cache, "w") as f:
        f.write(lines[0][: len(lines[0]) // 2] + "\n" + "spmvacc1 00ff 1 2 3\n" + "\x00\x01 not a record\n\n" + "\n".join(lines) + "\n")
    second, log2 = _run_child(tmp_path, "p2", strategies, {"SPMV_ACC_TUNE_CACHE": cache})
    assert "adopted from the tune cache" in log2
    assert "stream policy" not in log2 and "-> family" not in log2 and "flat cut rows" not in log2 and "encoding %" not in log2 and ": colindex " not in log2, log2[-2000:]
    for s in strategies:
        assert np.array_equal(first[s], second[s]), s
        # same timed choices (stream policy, adaptive's family); the second process built only the family that won, the first all three
        assert first[s + "__plan"][6] == second[s + "__plan"][6] and first[s + "__plan"][8] == second[s + "__plan"][8], s
    # (the first call of a fresh PROCESS also loads the kernels' code objects -- milliseconds, and not the same every time: 0.3 .. 7.6 ms seen for
    # the second process against 10 .. 13 for the first; the cheap first call with cached choices is measured inside one process by
    # tools/prepare_cost.py, profiles/r03_prepare_cost.txt: 0.34 ms against 9.9)
    assert float(second["adaptive__prepare_ms"]) < float(first["adaptive__prepare_ms"]), (first["adaptive__prepare_ms"], second["adaptive__prepare_ms"])


_CHILD_POWERLAW = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
import spmv_acc_amd
m, n, per_row = 300_000, 16_000_000, 40   # x = 128 MB: beyond the bound below which the column census is skipped
g = torch.Generator(device="cuda"); g.manual_seed(11)
nnz = m * per_row
rows = torch.arange(m, device="cuda").repeat_interleave(per_row)
ci = (torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) ** 6 * n).long().clamp_(0, n - 1)
ci = (torch.sort(rows * n + ci).values % n).to(torch.int32)   # ascending inside every row
rp = (torch.arange(m + 1, device="cuda") * per_row).to(torch.int32)
v = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
ms = spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="adaptive_plus")
y = torch.ones(m, dtype=torch.float64, device="cuda")
spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy="adaptive_plus")
torch.cuda.synchronize()
ref = 1.0 + torch.segment_reduce(v * x[ci.long()], "sum", lengths=torch.full((m,), per_row, device="cuda"), unsafe=True)
scale = 1.0 + torch.segment_reduce((v * x[ci.long()]).abs(), "sum", lengths=torch.full((m,), per_row, device="cuda"), unsafe=True)
np.savez({out!r}, y=y.cpu().numpy(), prepare_ms=np.array(ms), slab_passes=np.array(spmv_acc_amd.query_plan(rp, m)["slab_passes"]),
         err=np.array(float(((y - ref).abs() / scale).max().item())))
"""


def test_slab_pass_choice_is_timed_once_and_kept_by_the_tune_cache(torch_dev, tmp_path):
    """A matrix with power-law columns and an x beyond the caches: the first process takes the column census, builds the run lists and
    times the slab passes against the row-block-plus kernel (one tune-log line says which stays); the second process adopts the
    choice from the cache without timing anything, runs the same path, and computes bitwise the same y."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cache = str(tmp_path / "tune.txt")
    res = []
    for tag in ("p1", "p2"):
        out = str(tmp_path / f"{tag}.npz")
        env = dict(os.environ, SPMV_ACC_TUNE_LOG="1", SPMV_ACC_TUNE_CACHE=cache)
        r = subprocess.run([sys.executable, "-c", _CHILD_POWERLAW.format(root=root, out=out)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        res.append((np.load(out), r.stderr))
    (first, log1), (second, log2) = res
    assert "column census" in log1 and "column-slab passes over run lists" in log1, log1[-3000:]
    assert "adopted from the tune cache" in log2 and "column-slab passes over run lists" not in log2, log2[-3000:]
    assert float(first["err"]) <= SCALED_TOL and float(second["err"]) <= SCALED_TOL
    assert int(first["slab_passes"]) == int(second["slab_passes"])
    assert np.array_equal(first["y"], second["y"])
    assert float(second["prepare_ms"]) < float(first["prepare_ms"])


def test_deterministic_switch_is_bitwise_stable_across_processes(torch_dev, tmp_path):
    """SPMV_ACC_DETERMINISTIC=1: nothing is timed (no timing line in the tune log), every choice follows a rule on the matrix'
    shape, so two processes compute bitwise the same y with every strategy."""
    strategies = tuple(spmv_acc_amd.STRATEGIES)
    a, log_a = _run_child(tmp_path, "d1", strategies, {"SPMV_ACC_DETERMINISTIC": "1"})
    b, log_b = _run_child(tmp_path, "d2", strategies, {"SPMV_ACC_DETERMINISTIC": "1"})
    for log in (log_a, log_b):
        assert " us" not in log.replace("census", ""), log[-1500:]  # no timing was taken
    for s in strategies:
        assert np.array_equal(a[s], b[s]), s


# ---- opt-in column-slab blocking (tunable col_slabs) --------------------------------------------------------------------------
@pytest.mark.parametrize("kind,m,n,avg", [("powerlaw", 30000, 30000, 12), ("uniform", 20000, 50000, 7), ("empty_rows", 15000, 15000, 5)])
def test_col_slabs_opt_in_matches_the_oracle(torch_dev, oracle, hiplib, kind, m, n, avg):
    """Tunable col_slabs = S (off by default): the plan holds the matrix re-ordered into S column-range slabs and an SpMV is S
    consecutive SpMVs of the named strategy (first applies beta, the rest accumulate).  Every strategy, S = 2 / 8 / 16 (more slabs
    than some rows have non-zeros: empty slabs), in place and out of place, beta = 0, against the oracle; the derived plans go away
    with their parent; an in-place edit of the caller's structure is still noticed."""
    torch = torch_dev
    rowptr, cols, vals = synth.random_csr(m, n, avg, seed=33, kind=kind)
    if kind == "uniform":  # columns confined to the first third of [0, n): most slabs of 8 and 16 are empty
        cols = (cols % (n // 3)).astype(np.int32)
        for i in range(m):  # keep the rows sorted (not required by the kernels, tidy for the oracle's error scale)
            a, b = rowptr[i], rowptr[i + 1]
            order = np.argsort(cols[a:b], kind="stable")
            cols[a:b] = cols[a:b][order]
            vals[a:b] = vals[a:b][order]
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    try:
        for S in (2, 8, 16):
            hiplib.spmv_acc_set_tunable(b"col_slabs", S)
            for strat in ALL:
                for alpha, beta in ((1.0, 1.0), (0.5, -2.0), (2.0, 0.0)):
                    ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
                    spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy=strat, beta=beta)  # (settled: the bitwise comparison below)
                    y = dy0.clone()
                    spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y, strategy=strat)
                    y_in = dy0.clone()
                    y_out = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
                    spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y_out, strategy=strat, y_in=y_in)
                    torch.cuda.synchronize()
                    assert oracle.scaled_error(y.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (S, strat, alpha, beta)
                    assert torch.equal(y_out, y) and torch.equal(y_in, dy0), (S, strat, "out of place")
            assert hiplib.spmv_acc_cached_plans() > 1  # the parent + its slabs
            spmv_acc_amd.release_plans(drp)
            assert hiplib.spmv_acc_cached_plans() == 0, "slab plans outlived their parent"
        # a row shard handed over WITHOUT rebasing (rowptr[0] > 0, the nnz argument is then the end offset): the slabs hold the
        # shard's non-zeros only (an earlier form sized the last slab from the end offset and gathered through uninitialised columns)
        r0, r1 = m // 5, m - m // 7
        hiplib.spmv_acc_set_tunable(b"col_slabs", 3)
        sl = slice(r0, r1)
        sub_rp = (rowptr[r0:r1 + 1] - rowptr[r0]).astype(np.int32)
        sub = (sub_rp, cols[rowptr[r0]:rowptr[r1]], vals[rowptr[r0]:rowptr[r1]])
        for strat in ("flat", "line_enhance", "adaptive_plus"):
            y = dy0.clone()
            spmv_acc_amd.csr_spmv(1.0, 1.0, r1 - r0, n, int(rowptr[r1]), drp[r0:], dci, dv, dx, y[r0:], strategy=strat)
            torch.cuda.synchronize()
            got = y.cpu().numpy()
            assert np.array_equal(got[:r0], y0[:r0]) and np.array_equal(got[r1:], y0[r1:]), (strat, "wrote outside the shard")
            ref = oracle.host_spmv(1.0, 1.0, *sub, x, y0[sl])
            assert oracle.scaled_error(got[sl], ref, 1.0, 1.0, *sub, x, y0[sl]) <= SCALED_TOL, (strat, "unrebased shard")
        spmv_acc_amd.release_plans(drp[r0:])
        # values changed in place: the slabs hold a copy -- spmv_acc_refresh_values re-copies them (structure and plans stay)
        hiplib.spmv_acc_set_tunable(b"col_slabs", 4)
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")
        torch.cuda.synchronize()
        plans = hiplib.spmv_acc_cached_plans()
        vals2 = vals * -0.75 + 0.125
        dv.copy_(dev(torch, vals2))
        assert spmv_acc_amd.refresh_values(drp) == 1
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")
        torch.cuda.synchronize()
        ref2 = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals2, x, y0)
        assert oracle.scaled_error(y.cpu().numpy(), ref2, 1.0, 1.0, rowptr, cols, vals2, x, y0) <= SCALED_TOL
        assert hiplib.spmv_acc_cached_plans() == plans and hiplib.spmv_acc_last_error() == 0
        dv.copy_(dev(torch, vals))
        assert spmv_acc_amd.refresh_values(drp) == 1
        spmv_acc_amd.release_plans(drp)
        assert spmv_acc_amd.refresh_values(drp) == 0  # nothing to refresh
        # the caller rewrites the structure in place (same nnz) without a release: the parent's guard still fires
        hiplib.spmv_acc_set_tunable(b"col_slabs", 4)
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")
        torch.cuda.synchronize()
        lens = np.diff(rowptr)[::-1].copy()
        drp.copy_(dev(torch, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)))
        torch.cuda.synchronize()
        hiplib.spmv_acc_csr_spmv_strategy(spmv_acc_amd.strategy_id("adaptive"), 0, 1.0, 1.0, m, n, nnz, None, drp.data_ptr(),
                                          dci.data_ptr(), dv.data_ptr(), dx.data_ptr(), y.data_ptr())
        torch.cuda.synchronize()
        if not np.array_equal(lens, np.diff(rowptr)):
            assert hiplib.spmv_acc_last_error() == 2 and b"changed" in hiplib.spmv_acc_last_error_string()
    finally:
        hiplib.spmv_acc_clear_error()
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


# ---- column-slab blocking without a copy (tunable slab_segments, k_segment.hip) ------------------------------------------------
def _sorted_rows(rowptr, cols, vals):
    cols, vals = cols.copy(), vals.copy()
    for i in range(len(rowptr) - 1):
        a, b = rowptr[i], rowptr[i + 1]
        order = np.argsort(cols[a:b], kind="stable")
        cols[a:b] = cols[a:b][order]
        vals[a:b] = vals[a:b][order]
    return cols, vals


@pytest.mark.parametrize("kind,m,n,avg", [("powerlaw", 30000, 30000, 12), ("uniform", 20000, 50000, 7), ("empty_rows", 15000, 15000, 5),
                                          ("longrows", 2500, 60000, 30)])
def test_slab_segments_match_the_oracle(torch_dev, oracle, hiplib, kind, m, n, avg):
    """Tunable slab_segments = S: where every row's columns ascend the plan keeps per column slab the list of (row, first non-zero,
    length) runs -- no copy of the matrix -- and an SpMV is S passes over those runs.  S = 2 / 8 / 16, general alpha / beta, in place
    and out of place, runs far longer than a piece (cut into pieces whose sums the merge kernel adds in order), empty rows, empty
    slabs, an un-rebased row shard; VALUES edited in place are seen by the next call without any refresh; COLUMNS edited in place (the
    runs are then filed under the wrong slabs) still give the right sums; rows that are not ordered take the ordinary path."""
    torch = torch_dev
    if kind == "longrows":
        rowptr, cols, vals = synth.random_csr(m, n, avg, seed=9, kind="uniform")
        rng = np.random.default_rng(5)
        lens = np.diff(rowptr).astype(np.int64)
        lens[[3, m // 2, m - 1]] = (20001, 5000, 1300)
        rowptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        cols = rng.integers(0, n, int(rowptr[-1])).astype(np.int32)
        vals = rng.standard_normal(int(rowptr[-1]))
    else:
        rowptr, cols, vals = synth.random_csr(m, n, avg, seed=33, kind=kind)
        if kind == "uniform":
            cols = (cols % (n // 3)).astype(np.int32)  # most slabs of 8 and 16 are empty
    cols, vals = _sorted_rows(rowptr, cols, vals)
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(3)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    try:
        for S in (2, 8, 16):
            hiplib.spmv_acc_set_tunable(b"slab_segments", S)
            for strat in ("adaptive", "flat"):  # (the passes replace whatever kernel the name would run)
                for alpha, beta in ((1.0, 1.0), (0.5, -2.0), (2.0, 0.0)):
                    ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
                    y = dy0.clone()
                    spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y, strategy=strat)
                    y_in = dy0.clone()
                    y_out = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
                    spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y_out, strategy=strat, y_in=y_in)
                    torch.cuda.synchronize()
                    assert oracle.scaled_error(y.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (S, strat, alpha, beta)
                    assert torch.equal(y_out, y) and torch.equal(y_in, dy0), (S, strat, "out of place")
            assert hiplib.spmv_acc_cached_plans() == 1  # no derived matrices, no derived plans
            spmv_acc_amd.release_plans(drp)
        hiplib.spmv_acc_set_tunable(b"slab_segments", 4)
        # a row shard handed over without rebasing (rowptr[0] > 0; the nnz argument is then the end offset)
        r0, r1 = m // 5, m - m // 7
        sl = slice(r0, r1)
        sub_rp = (rowptr[r0:r1 + 1] - rowptr[r0]).astype(np.int32)
        sub = (sub_rp, cols[rowptr[r0]:rowptr[r1]], vals[rowptr[r0]:rowptr[r1]])
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, r1 - r0, n, int(rowptr[r1]), drp[r0:], dci, dv, dx, y[r0:], strategy="line_enhance")
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        assert np.array_equal(got[:r0], y0[:r0]) and np.array_equal(got[r1:], y0[r1:]), "wrote outside the shard"
        ref = oracle.host_spmv(1.0, 1.0, *sub, x, y0[sl])
        assert oracle.scaled_error(got[sl], ref, 1.0, 1.0, *sub, x, y0[sl]) <= SCALED_TOL, "unrebased shard"
        spmv_acc_amd.release_plans(drp[r0:])
        # values edited in place: nothing to refresh, the plan holds no values
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")
        torch.cuda.synchronize()
        vals2 = vals * -0.75 + 0.125
        dv.copy_(dev(torch, vals2))
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")
        torch.cuda.synchronize()
        ref2 = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals2, x, y0)
        assert oracle.scaled_error(y.cpu().numpy(), ref2, 1.0, 1.0, rowptr, cols, vals2, x, y0) <= SCALED_TOL
        # columns edited in place, rows no longer ordered: the lists (built for the old columns) still cut every row into the same
        # runs, so the sums are over the same non-zeros -- only the locality the slabs were for is gone
        cols3 = ((cols.astype(np.int64) * 7919 + 13) % n).astype(np.int32)
        dci.copy_(dev(torch, cols3))
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")
        torch.cuda.synchronize()
        ref3 = oracle.host_spmv(1.0, 1.0, rowptr, cols3, vals2, x, y0)
        assert oracle.scaled_error(y.cpu().numpy(), ref3, 1.0, 1.0, rowptr, cols3, vals2, x, y0) <= SCALED_TOL
        assert hiplib.spmv_acc_last_error() == 0 and hiplib.spmv_acc_cached_plans() == 1
        # a FRESH plan on the unordered columns: no lists, the ordinary path
        spmv_acc_amd.release_plans(drp)
        y = dy0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")
        torch.cuda.synchronize()
        assert oracle.scaled_error(y.cpu().numpy(), ref3, 1.0, 1.0, rowptr, cols3, vals2, x, y0) <= SCALED_TOL
        # the structure rewritten in place without a release: the guard check ahead of the passes fires
        lens = np.diff(rowptr)[::-1].copy()
        if not np.array_equal(lens, np.diff(rowptr)):
            dci.copy_(dev(torch, cols))
            spmv_acc_amd.release_plans(drp)
            y = dy0.clone()
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy="adaptive")
            torch.cuda.synchronize()
            drp.copy_(dev(torch, np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)))
            torch.cuda.synchronize()
            hiplib.spmv_acc_csr_spmv_strategy(spmv_acc_amd.strategy_id("adaptive"), 0, 1.0, 1.0, m, n, nnz, None, drp.data_ptr(),
                                              dci.data_ptr(), dv.data_ptr(), dx.data_ptr(), y.data_ptr())
            torch.cuda.synchronize()
            assert hiplib.spmv_acc_last_error() == 2 and b"changed" in hiplib.spmv_acc_last_error_string()
    finally:
        hiplib.spmv_acc_clear_error()
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


@pytest.mark.parametrize("whole_below", [0, 2, 8, 32, 1 << 30])
def test_slab_passes_two_class_form(torch_dev, oracle, hiplib, whole_below):
    """Round 4, tunable slab_whole_below = T: rows of fewer than T non-zeros are not cut by column slab -- each is ONE run, all columns, in a
    pass of its own behind the S slab passes (T = 0: round 3's form, every row cut; T huge: every row whole, the slab passes are empty).  Same
    sums whatever T (the oracle's, to the scaled tolerance), bitwise equal between two runs, the short rows need no column order, and
    query_plan reports the COLUMN slabs."""
    torch = torch_dev
    m, n = 50000, 80000
    rowptr, cols, vals = synth.random_csr(m, n, 14, seed=21, kind="powerlaw")
    lens = np.diff(rowptr)
    cols, vals = _sorted_rows(rowptr, cols, vals)
    if whole_below > 2:  # rows below the threshold may be unsorted: shuffle them
        rng = np.random.default_rng(2)
        for i in np.nonzero((lens > 1) & (lens < min(whole_below, 1 << 20)))[0][:4000]:
            a, b = rowptr[i], rowptr[i + 1]
            perm = rng.permutation(b - a)
            cols[a:b], vals[a:b] = cols[a:b][perm], vals[a:b][perm]
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(8)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    try:
        hiplib.spmv_acc_set_tunable(b"slab_segments", 8)
        hiplib.spmv_acc_set_tunable(b"slab_whole_below", whole_below)
        for alpha, beta in ((1.0, 1.0), (-0.5, 0.0), (2.0, 3.0)):
            ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
            y = dy0.clone()
            spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y, strategy="line_enhance")
            y2 = dy0.clone()
            spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y2, strategy="line_enhance")
            torch.cuda.synchronize()
            assert oracle.scaled_error(y.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (whole_below, alpha, beta)
            assert torch.equal(y, y2)
        if whole_below <= 2 or (whole_below >= (1 << 20)) or int((lens >= whole_below).sum()) > 0:
            assert spmv_acc_amd.query_plan(drp, m)["slab_passes"] == 8
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


@pytest.mark.parametrize("slabs,whole_below", [(16, 32), (16, 0), (15, 32), (40, 32)])
def test_slab_planes_never_exceed_what_the_count_kernels_hold(torch_dev, oracle, hiplib, slabs, whole_below):
    """Regression (round 4, found by profiles/probes/rmat26_check.py on R-MAT 26): 16 column slabs -- what the automatic mode picks once x reaches 496 MB --
    plus the whole-row plane of the two-class form made 17 planes, one more than the count kernels keep counters for; rows of EXACTLY 32 non-zeros (the
    only ones the one-lane count kernel cuts by slab) then had one slab's run filed twice.  The build now gives the whole-row plane one of the 16; the
    matrix here is mostly such rows, spread over all columns."""
    torch = torch_dev
    m, n = 40000, 90000
    rng = np.random.default_rng(77)
    lens = rng.choice([32, 32, 32, 31, 33, 7, 64, 200], size=m)
    rowptr, cols, vals = synth.csr_from_row_lengths(lens, n, rng, locality=20000, far_fraction=0.3)
    cols, vals = _sorted_rows(rowptr, cols, vals)
    nnz = int(rowptr[-1])
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    try:
        hiplib.spmv_acc_set_tunable(b"slab_segments", slabs)
        hiplib.spmv_acc_set_tunable(b"slab_whole_below", whole_below)
        for strat in ("line_enhance", "flat"):
            ref = oracle.host_spmv(0.5, -2.0, rowptr, cols, vals, x, y0)
            y = dy0.clone()
            spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, y, strategy=strat)
            torch.cuda.synchronize()
            assert oracle.scaled_error(y.cpu().numpy(), ref, 0.5, -2.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, (strat, slabs, whole_below)
            assert spmv_acc_amd.query_plan(drp, m)["slab_passes"] == min(slabs, 16 - (1 if whole_below > 1 else 0))
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


@pytest.mark.parametrize("deterministic", [1, 0])
def test_whole_row_pass_with_gather_hints(torch_dev, oracle, hiplib, deterministic):
    """Round 4: the whole-row pass of the slab lists (rows below slab_whole_below) takes the plan's gather hints -- cold gathers through the raw-buffer
    non-temporal path -- where a timing of that pass alone says they pay (on an x far larger than the caches: R-MAT 25 1.14 -> 1.02 ms).  At test
    size nothing pays, so `gather_hint = 1` builds the hints whatever n is and `deterministic` takes the rule (hinted) instead of the timing; without
    it the timed choice runs, whichever way it falls.  Same sums either way, to the oracle's tolerance, for in-place and out-of-place calls."""
    torch = torch_dev
    m, n = 60000, 200000
    rowptr, cols, vals = synth.random_csr(m, n, 9, seed=33, kind="powerlaw")
    cols, vals = _sorted_rows(rowptr, cols, vals)
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(12)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    try:
        for k, val in (("slab_segments", 4), ("gather_hint", 1), ("deterministic", deterministic)):
            assert hiplib.spmv_acc_set_tunable(k.encode(), val) == 0
        for alpha, beta in ((1.0, 1.0), (-0.5, 0.0), (2.0, 3.0)):
            ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
            y = dy0.clone()
            spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y, strategy="line_enhance")
            y_in, y_out = dy0.clone(), torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
            spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y_out, strategy="line_enhance", y_in=y_in)
            torch.cuda.synchronize()
            assert oracle.scaled_error(y.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (alpha, beta)
            assert oracle.scaled_error(y_out.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (alpha, beta, "out of place")
        assert spmv_acc_amd.query_plan(drp, m)["slab_passes"] == 4
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


def test_slab_segments_replay_from_a_graph_and_are_bitwise_stable(torch_dev, oracle, hiplib):
    """The passes are ordinary launches over plan-resident lists: captured after one warm-up call they replay, and two runs give the
    same bits (whole runs add straight into y, the pieces of a long run are added in entry order by one thread: no atomics)."""
    torch = torch_dev
    m = n = 40000
    rowptr, cols, vals = synth.random_csr(m, n, 10, seed=12, kind="powerlaw")
    cols, vals = _sorted_rows(rowptr, cols, vals)
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(6)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    y = dy0.clone()
    stream = torch.cuda.Stream()
    try:
        hiplib.spmv_acc_set_tunable(b"slab_segments", 8)
        sid = spmv_acc_amd.strategy_id("line_enhance")
        with torch.cuda.stream(stream):
            hiplib.spmv_acc_set_stream(stream.cuda_stream)
            spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, drp, dci, dv, dx, y, strategy="line_enhance")
            stream.synchronize()
            first = y.clone()
            y.copy_(dy0)
            stream.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                hiplib.spmv_acc_csr_spmv_strategy(sid, 0, 0.5, -2.0, m, n, nnz, None, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(),
                                                  dx.data_ptr(), y.data_ptr())
            assert hiplib.spmv_acc_last_error() == 0, hiplib.spmv_acc_last_error_string()
        ref = oracle.host_spmv(0.5, -2.0, rowptr, cols, vals, x, y0)
        for _ in range(3):
            y.copy_(dy0)
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(y, first)
        assert oracle.scaled_error(y.cpu().numpy(), ref, 0.5, -2.0, rowptr, cols, vals, x, y0) <= SCALED_TOL
    finally:
        hiplib.spmv_acc_set_stream(None)
        hiplib.spmv_acc_clear_error()
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


# ---- genuine LIGHT and BLOCK_ROW_ORDINARY kernels (the last two KERNEL_STRATEGY names that were aliases) -----------------------------
def test_light_and_block_row_run_their_own_kernels(torch_dev, oracle, hiplib):
    """KERNEL_STRATEGY=LIGHT hands rows out through an atomic counter (LightSpMV, hip-light/spmv_hip_acc_imp.inl:36-76), BLOCK_ROW_ORDINARY
    gives every row a whole workgroup (hip-block-row-ordinary/spmv_hip_acc_imp.cpp:16-66).  Both against the oracle on matrices that
    exercise their corners -- every lane width of LIGHT (average row length 1 .. 200), rows far longer than a workgroup's step, empty
    rows, a row count that is not a multiple of the fetch size, fewer rows than the resident grid -- with general alpha / beta, out of
    place, inside a hipGraph (LIGHT's counter is left at zero by the kernel's last wavefront: one node), and bitwise reproducible although LIGHT's row-to-wave
    assignment changes from launch to launch.  (The round-2 stand-ins behind `legacy_kernels = 0` went in round 6.)"""
    torch = torch_dev
    cases = []
    for avg, m in ((1, 50_001), (3, 40_000), (7, 30_011), (14, 20_000), (30, 9_000), (60, 5_000), (200, 1_500)):
        cases.append((f"avg {avg}", synth.random_csr(m, m, avg, seed=avg, kind="powerlaw" if avg > 3 else "uniform")))
    cases.append(("empty rows", synth.random_csr(20_000, 20_000, 6, seed=4, kind="empty_rows")))
    cases.append(("spikes", synth.random_csr(3_000, 40_000, 9, seed=5, kind="spikes")))
    cases.append(("tiny", synth.random_csr(37, 50, 4, seed=6, kind="uniform")))
    rng = np.random.default_rng(2)
    side = torch.cuda.Stream()
    try:
        for tag, (rowptr, cols, vals) in cases:
            m, n, nnz = rowptr.size - 1, int(cols.max()) + 1 if cols.size else 1, int(rowptr[-1])
            x, y0 = rng.standard_normal(n), rng.standard_normal(m)
            drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
            for strat in ("light", "block_row_ordinary"):
                for alpha, beta in ((1.0, 1.0), (0.5, -2.0), (2.0, 0.0)):
                    ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
                    y = dy0.clone()
                    spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y, strategy=strat)
                    y2 = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
                    spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, drp, dci, dv, dx, y2, strategy=strat, y_in=dy0)
                    torch.cuda.synchronize()
                    assert oracle.scaled_error(y.cpu().numpy(), ref, alpha, beta, rowptr, cols, vals, x, y0) <= SCALED_TOL, (tag, strat, alpha, beta)
                    assert torch.equal(y, y2), (tag, strat, "out of place / run-to-run")  # same sums whichever wave took which rows
                # captured and replayed
                with torch.cuda.stream(side):
                    static_y = dy0.clone()
                    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, static_y, strategy=strat)
                side.synchronize()
                eager = static_y.clone()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=side):
                    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, static_y, strategy=strat)
                for _ in range(2):
                    static_y.copy_(dy0)
                    g.replay()
                    torch.cuda.synchronize()
                    assert torch.equal(static_y, eager), (tag, strat, "graph replay")
            hiplib.spmv_acc_reset_tunables()
            spmv_acc_amd.release_plans(drp)
    finally:
        hiplib.spmv_acc_set_stream(None)
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans()


# ---- bench.py --gpus N on the GPU box: the self-launch and the N > 1 code path, two gloo ranks sharing the card ---------------------
def test_bench_gpus_2_launches_itself_and_runs_the_sharded_step(torch_dev):
    """`python bench.py --gpus 2` with no WORLD_SIZE -- what the driver runs when it has more than one GPU -- on this one-GPU box:
    SPMV_ACC_BENCH_BACKEND=gloo lets the two ranks share the card (RCCL needs a GPU per rank).  The parent launches the ranks as
    child processes, rank 0's ONE JSON line comes back with n_gpus 2, weak scaling (value counts both ranks' non-zeros), the
    spmv_only / spmv+exchange split and the exchange form; exit status 0."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPMV_ACC_BENCH_CHILD")}
    env["SPMV_ACC_BENCH_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--scale", "0.1", "--no-legs"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:500]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 10
    assert "x2" in d["config"]["parallelism"] and "gloo" in d["config"]["parallelism"]
    assert d["exchange"] in ("allgather", "p2p") and d["spmv_plus_exchange_ms_per_step"] > 0 and d["spmv_only_gflops_per_gpu"] > 0
    assert d["allgather_bytes_per_rank_per_step"] == 8 * d["config"]["rows_per_gpu"]  # one peer's slice
    # value = both ranks' non-zeros over the max-over-ranks wall time
    assert abs(d["value"] - 2.0 * 2 * d["config"]["nnz_per_gpu"] * d["steps"] / (d["ms_per_step"] * 1e-3 * d["steps"]) / 1e9) / d["value"] < 0.02
    assert "launching 2 ranks" in r.stderr


# ---- round 4: the first call's tuning budget ----------------------------------------------------------------------------------------
def test_first_call_is_bounded_and_later_calls_finish_the_timings(torch_dev, oracle, hiplib):
    """Tunables first_call_budget / later_call_budget (round 4): the FIRST call on a matrix spends at most ~20 SpMV-equivalents on trial
    launches, leaves the choices it did not get to OPEN (the `deterministic` rule serves meanwhile) and the following calls finish them, one
    phase each; spmv_acc_prepare pays for everything up front.  Every call is right whatever the state of the plan; the bounded first
    call does less plan work than the unbounded one (first_call_budget = 0: rounds 1-3); after a few calls nothing is left to do and the plan
    reports its choices; a prepared plan has nothing left for the calls behind it."""
    torch = torch_dev
    m = n = 1_500_000
    rowptr, cols, vals = synth.random_csr(m, n, 9, seed=77, kind="uniform")
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(12)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    lib = hiplib

    def calls(k, strat):
        work = []
        for _ in range(k):
            y = dy0.clone()
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y, strategy=strat)
            torch.cuda.synchronize()
            work.append(lib.spmv_acc_last_prepare_us())
            assert oracle.scaled_error(y.cpu().numpy(), ref, 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL, (strat, len(work))
        return work

    try:
        for strat in ("adaptive", "flat"):
            spmv_acc_amd.release_plans(drp)
            calls(1, strat)  # (this process's first use of the kernels: code-object loads are not the plan's cost)
            spmv_acc_amd.release_plans(drp)
            lib.spmv_acc_set_tunable(b"first_call_budget", 0)  # unbounded: everything on the call's path in call 1
            unbounded = calls(3, strat)
            lib.spmv_acc_reset_tunables()
            spmv_acc_amd.release_plans(drp)
            first = calls(1, strat)
            settled_after_first = spmv_acc_amd.query_plan(drp, m)["settled"]
            bounded = first + calls(13, strat)
            assert settled_after_first == (sum(1 for w in bounded[1:] if w > 0) == 0), (strat, bounded)  # open timings <=> later plan work
            assert bounded[0] > 0 and unbounded[0] > 0
            assert bounded[0] < 0.8 * unbounded[0], (strat, bounded[0], unbounded[0])  # the first call does less ...
            assert sum(1 for w in bounded[1:] if w > 0) >= 1, (strat, bounded)          # ... later calls do the rest ...
            assert bounded[-1] == 0.0 and bounded[-2] == 0.0, (strat, bounded)          # ... and then it is over
            info = spmv_acc_amd.query_plan(drp, m)
            assert info["stream_policy"] in (0, 1, 3) and info["settled"]  # (spmv_acc_query_plan_settled: nothing left open)
            if strat == "adaptive":
                assert info["adaptive_family"] in (0, 1, 2) and info["flat_tiles"] > 0 and info["plus_blocks"] > 0  # every family was looked at
            # everything up front instead
            spmv_acc_amd.release_plans(drp)
            assert spmv_acc_amd.prepare(m, n, nnz, drp, dci, dv, dx, strategy=strat) > 0
            assert calls(3, strat) == [0.0, 0.0, 0.0], strat
            # and a settled plan gives the same bits call after call
            y1, y2 = dy0.clone(), dy0.clone()
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y1, strategy=strat)
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, drp, dci, dv, dx, y2, strategy=strat)
            torch.cuda.synchronize()
            assert torch.equal(y1, y2)
    finally:
        lib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans(drp)


def test_chunks_entry_equals_per_chunk_calls(torch_dev, oracle, hiplib):
    """spmv_acc_csr_spmv_chunks (round 4): row sub-ranges of one matrix as consecutive launches alternating over two streams, an event behind
    each, in ONE host call -- the compute side of the pipelined sharded step.  Same bits as one call per chunk on un-rebased views; rows outside
    the cuts untouched; every event is recorded behind its chunk; the calling thread's library stream is left alone; ragged and empty chunks."""
    import ctypes

    torch = torch_dev
    m, n = 120_000, 90_000
    rowptr, cols, vals = synth.random_csr(m, n, 8, seed=5, kind="powerlaw")
    nnz = int(rowptr[-1])
    rng = np.random.default_rng(2)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    drp, dci, dv, dx, dy0 = (dev(torch, a) for a in (rowptr, cols, vals, x, y0))
    cuts = [1000, 1000, 31000, 31007, 90000, 119999]  # an empty chunk, a 7-row chunk, rows [0, 1000) and the last row left out
    ends = [int(rowptr[c]) for c in cuts[1:]]
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    evs = []
    for k in range(len(cuts) - 1):
        e = torch.cuda.Event()
        e.record((s0, s1)[k & 1])
        evs.append(e)
    torch.cuda.synchronize()
    before = hiplib.spmv_acc_get_stream()
    try:
        for strat in ("adaptive", "flat", "line_enhance"):
            for alpha, beta in ((1.0, 1.0), (0.5, 0.0)):
                # per-chunk reference on settled plans
                want = dy0.clone()
                for k in range(len(cuts) - 1):
                    a, b = cuts[k], cuts[k + 1]
                    if b > a:
                        spmv_acc_amd.prepare(b - a, n, ends[k], drp[a:], dci, dv, dx, strategy=strat, beta=beta)
                        spmv_acc_amd.csr_spmv(alpha, beta, b - a, n, ends[k], drp[a:], dci, dv, dx, want[a:], strategy=strat)
                torch.cuda.synchronize()
                got = dy0.clone()
                torch.cuda.synchronize()
                rc = hiplib.spmv_acc_csr_spmv_chunks(spmv_acc_amd.strategy_id(strat), alpha, beta, n, len(cuts) - 1, (ctypes.c_int * len(cuts))(*cuts),
                                                     (ctypes.c_int * len(ends))(*ends), drp.data_ptr(), dci.data_ptr(), dv.data_ptr(), dx.data_ptr(), 0,
                                                     got.data_ptr(), (ctypes.c_void_p * 2)(s0.cuda_stream, s1.cuda_stream),
                                                     (ctypes.c_void_p * len(evs))(*[int(e.cuda_event) for e in evs]))
                assert rc == 0, hiplib.spmv_acc_last_error_string()
                for e in evs:
                    e.synchronize()  # (each event sits behind its chunk's kernels: waiting for all of them is waiting for the step)
                assert torch.equal(got, want), (strat, alpha, beta)
                assert torch.equal(got[:1000], dy0[:1000]) and torch.equal(got[119999:], dy0[119999:])
                assert hiplib.spmv_acc_get_stream() == before
        ref = oracle.host_spmv(1.0, 1.0, rowptr, cols, vals, x, y0)
        y = dy0.clone()
        full_cuts = [0, 40000, 80000, m]
        torch.cuda.synchronize()
        rc = hiplib.spmv_acc_csr_spmv_chunks(-1, 1.0, 1.0, n, 3, (ctypes.c_int * 4)(*full_cuts), (ctypes.c_int * 3)(*[int(rowptr[c]) for c in full_cuts[1:]]),
                                             drp.data_ptr(), dci.data_ptr(), dv.data_ptr(), dx.data_ptr(), 0, y.data_ptr(),
                                             (ctypes.c_void_p * 2)(s0.cuda_stream, s1.cuda_stream), None)
        torch.cuda.synchronize()
        assert rc == 0 and oracle.scaled_error(y.cpu().numpy(), ref, 1.0, 1.0, rowptr, cols, vals, x, y0) <= SCALED_TOL
        assert hiplib.spmv_acc_csr_spmv_chunks(-1, 1.0, 1.0, n, 2, None, None, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(), dx.data_ptr(), 0, y.data_ptr(),
                                               None, None) == 2  # SPMV_ACC_ERR_BAD_ARGUMENT
    finally:
        hiplib.spmv_acc_clear_error()
        spmv_acc_amd.release_plans()
