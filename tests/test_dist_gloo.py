"""CPU suite, world_size 2 and 3 (gloo): the row-range shard + allgather(y) path in both exchange forms.  The local SpMV is injected
(the oracle stands in for the HIP kernels, which need a GPU); partition, CSR slicing, padding, the
double-buffered allgather and the re-assembly of y are the product code under test."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_local_spmv(oracle_lib):
    """The oracle stands in for the HIP kernels on CPU, behind the engine's local-compute interface:
    local_spmv(alpha, beta, x, y_out, y_in, chunk) with chunk = (row0, row1, rowptr, cols, vals, nnz, host_rowptr) of the rows
    to compute (rebased CSR of that row range) and y_out / y_in views of those rows; y_in None = beta is 0."""
    def local_spmv(alpha, beta, xt, y_out, y_in, chunk):
        a, b, rp, ci, v = chunk[:5]
        tmp = y_in.numpy().copy() if y_in is not None else np.zeros(b - a)
        oracle_lib.host_spmv_inplace(alpha, beta, np.ascontiguousarray(rp), np.ascontiguousarray(ci), np.ascontiguousarray(v),
                                     xt.numpy(), tmp)
        y_out.numpy()[:] = tmp
    return local_spmv


def _worker(rank, world, port, mode, m, out_path, exchange="allgather", pipeline=1):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from spmv_acc_amd import synth
    from spmv_acc_amd.dist import RowShardedSpmv, local_csr_slice, shard_bounds

    n = m
    rowptr, cols, vals = synth.random_csr(m, n, 7, seed=123, kind="powerlaw")  # same matrix on every rank
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n)
    y0 = rng.standard_normal(m)
    bounds = shard_bounds(m, world, mode=mode, h_rowptr=rowptr)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    rp, ci, v = local_csr_slice(rowptr, cols, vals, r0, r1)
    rp, ci, v = np.ascontiguousarray(rp), np.ascontiguousarray(ci), np.ascontiguousarray(v)

    local_spmv = _oracle_local_spmv(oracle_lib)
    eng = RowShardedSpmv(rank, world, bounds, rp, ci, v, n, torch.device("cpu"), local_spmv=local_spmv,
                         exchange="allgather" if exchange == "tune" else exchange, pipeline=pipeline)
    tuned = eng.tune_exchange(warm=1, iters=2) if exchange == "tune" else None
    xt = torch.from_numpy(x)
    ylocal = torch.from_numpy(y0[r0:r1].copy())
    results = []
    for alpha, beta in ((1.0, 1.0), (0.5, 0.0), (2.0, -1.0)):
        eng.step(alpha, beta, xt, y_prev=ylocal, overlap=True)
        eng.step(alpha, beta, xt, y_prev=ylocal, overlap=True)  # second step exercises the two alternating vectors
        results.append(eng.gathered().numpy().copy())
        assert torch.equal(ylocal, torch.from_numpy(y0[r0:r1])), "y_prev was written"
    if rank == 0:
        np.savez(out_path, bounds=bounds, exchange=np.array(eng.exchange), tuned=np.array(sorted(tuned) if tuned else []),
                 **{f"y{i}": r for i, r in enumerate(results)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,m,world,exchange,pipeline", [(0, 4001, 2, "allgather", 1), (1, 4001, 2, "allgather", 1), (0, 4096, 2, "allgather", 1),
                                                            (0, 4001, 2, "p2p", 1), (1, 4001, 3, "p2p", 1), (0, 4099, 3, "tune", 1),
                                                            (0, 4001, 2, "allgather", 2), (1, 4001, 3, "p2p", 3), (1, 4099, 2, "allgather", 5),
                                                            (0, 4097, 3, "p2p", 8)])
def test_row_sharded_spmv_world2(tmp_path, oracle, mode, m, world, exchange, pipeline):
    """world 2 and 3, both exchange forms (RCCL allgather / direct fan-out), the timed choice between them, and the pipelined
    step (the local rows cut into 2 / 3 / 5 / 8 chunks, each chunk's slice travelling point to point as soon as it is computed,
    shards of unequal row counts and chunk sizes that do not divide them): every variant equals the unsharded oracle bit for bit."""
    from spmv_acc_amd import synth

    out = str(tmp_path / "y.npz")
    mp.spawn(_worker, args=(world, _free_port(), mode, m, out, exchange, pipeline), nprocs=world, join=True)
    g = np.load(out)
    if exchange == "tune":
        assert list(g["tuned"]) == ["allgather", "p2p"] and str(g["exchange"]) in ("allgather", "p2p")
    else:
        assert str(g["exchange"]) == exchange
    rowptr, cols, vals = synth.random_csr(m, m, 7, seed=123, kind="powerlaw")
    rng = np.random.default_rng(5)
    x = rng.standard_normal(m)
    y0 = rng.standard_normal(m)
    b = g["bounds"]
    assert b[0] == 0 and b[-1] == m and len(b) == world + 1
    for i, (alpha, beta) in enumerate(((1.0, 1.0), (0.5, 0.0), (2.0, -1.0))):
        ref = oracle.host_spmv(alpha, beta, rowptr, cols, vals, x, y0)
        assert g[f"y{i}"].shape == (m,)
        assert np.array_equal(g[f"y{i}"], ref), (mode, alpha, beta)  # same arithmetic per row: bit-exact


def _inplace_worker(rank, world, port, m, steps, out_path, pipeline=1, dependent=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from spmv_acc_amd import synth
    from spmv_acc_amd.dist import RowShardedSpmv, local_csr_slice, shard_bounds

    rowptr, cols, vals = synth.random_csr(m, m, 6, seed=77, kind="uniform")
    rng = np.random.default_rng(9)
    x, y0 = rng.standard_normal(m), rng.standard_normal(m)
    bounds = shard_bounds(m, world, mode=0)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    rp, ci, v = (np.ascontiguousarray(a) for a in local_csr_slice(rowptr, cols, vals, r0, r1))

    eng = RowShardedSpmv(rank, world, bounds, rp, ci, v, m, torch.device("cpu"), local_spmv=_oracle_local_spmv(oracle_lib),
                         pipeline=pipeline)
    eng.set_y(torch.from_numpy(y0[r0:r1].copy()))
    xt = torch.from_numpy(x)
    ys = []
    for _ in range(steps):  # y <- 0.25 * A x + 0.5 * y, no y_prev: every step must read the y of the step before it
        eng.step(0.25, 0.5, xt, overlap=not dependent)
        ys.append(eng.gathered().numpy().copy())
        if dependent:  # x_{k+1} = the gathered y_k (square matrix): the vector the step just completed IS the next x
            assert eng.y_full.numel() >= m
            xt = eng.gathered().clone()
    if rank == 0:
        np.savez(out_path, **{f"y{i}": y for i, y in enumerate(ys)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pipeline,dependent", [(2, 1, False), (3, 1, False), (2, 4, False), (3, 2, True), (2, 3, True)])
def test_row_sharded_in_place_iteration_follows_the_serial_recurrence(tmp_path, oracle, world, pipeline, dependent):
    """beta != 0 without y_prev iterates in place: step k reads the y step k-1 wrote although the two gathered vectors alternate
    (an earlier form read the y of step k-2) -- unpipelined, pipelined, and as the DEPENDENT iteration x_{k+1} = gathered y_k
    that the pipelined step exists for."""
    from spmv_acc_amd import synth

    m, steps = 3001, 4
    out = str(tmp_path / "y.npz")
    mp.spawn(_inplace_worker, args=(world, _free_port(), m, steps, out, pipeline, dependent), nprocs=world, join=True)
    g = np.load(out)
    rowptr, cols, vals = synth.random_csr(m, m, 6, seed=77, kind="uniform")
    rng = np.random.default_rng(9)
    x, y = rng.standard_normal(m), rng.standard_normal(m)
    for i in range(steps):
        y = oracle.host_spmv(0.25, 0.5, rowptr, cols, vals, x, y)
        assert np.array_equal(g[f"y{i}"], y), i
        if dependent:
            x = y.copy()


def test_shard_helpers():
    from spmv_acc_amd import synth
    from spmv_acc_amd.dist import local_csr_slice, padded_shard_rows, shard_bounds

    rowptr, cols, vals = synth.random_csr(1000, 1000, 5, seed=3)
    b = shard_bounds(1000, 8, mode=0)
    assert padded_shard_rows(b) == 125
    b = shard_bounds(1003, 8, mode=0)
    assert padded_shard_rows(b) == 126 and b[-1] == 1003
    rp, ci, v = local_csr_slice(rowptr, cols, vals, 100, 300)
    assert rp[0] == 0 and rp.size == 201 and ci.size == rp[-1] == rowptr[300] - rowptr[100]
    assert np.array_equal(ci, cols[rowptr[100]:rowptr[300]])


def _ghost_worker(rank, world, port, kind, m, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib
    from spmv_acc_amd.dist import GhostedRowShardedSpmv, local_csr_slice, shard_bounds

    rowptr, cols, vals, x0 = _ghost_problem(kind, m)
    bounds = shard_bounds(m, world, mode=1 if kind == "powerlaw" else 0, h_rowptr=rowptr)
    r0, r1 = int(bounds[rank]), int(bounds[rank + 1])
    rp, ci, v = (np.ascontiguousarray(a) for a in local_csr_slice(rowptr, cols, vals, r0, r1))
    seen = {}

    def local_spmv(alpha, beta, xt, yt):  # oracle stands in for the HIP kernel on CPU; cols are the engine's local numbering
        seen["n"] = xt.numel()
        oracle_lib.host_spmv_inplace(alpha, beta, rp, eng.cols_local.numpy(), v, xt.numpy(), yt.numpy())

    eng = GhostedRowShardedSpmv(rank, world, bounds, torch.from_numpy(rp), torch.from_numpy(ci), torch.from_numpy(v),
                                torch.device("cpu"), local_spmv=local_spmv)
    x = torch.from_numpy(x0[r0:r1].copy())
    y = torch.zeros(r1 - r0, dtype=torch.float64)
    for _ in range(3):  # x_{k+1} = 0.5 * A x_k + 0.25 * x_k, every rank holding only its slice
        eng.set_x(x)
        y.copy_(x)
        eng.step(0.5, 0.25, y)
        x = y.clone()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), x=x.numpy(), r0=r0, r1=r1, n_ghost=eng.n_ghost, n_ext=seen.get("n", 0),
             bytes=eng.exchanged_bytes_per_step)
    dist.barrier()
    dist.destroy_process_group()


def _ghost_problem(kind, m):
    from spmv_acc_amd import synth

    if kind == "banded":
        offs = np.arange(-4, 4)
        rows = np.arange(m)[:, None] + offs[None, :]
        ok = (rows >= 0) & (rows < m)
        rowptr = np.concatenate([[0], np.cumsum(ok.sum(1))]).astype(np.int32)
        cols = rows[ok].astype(np.int32)
        vals = (np.where(rows % 2 == 0, 1.0, -1.0) / (1.0 + np.abs(offs))[None, :])[ok]
    else:
        rowptr, cols, vals = synth.random_csr(m, m, 7, seed=321, kind=kind)
    x0 = np.random.default_rng(8).standard_normal(m)
    return rowptr, cols, vals, x0


@pytest.mark.parametrize("kind,m,world", [("banded", 3001, 2), ("banded", 3001, 3), ("powerlaw", 2500, 2), ("uniform", 1999, 3),
                                          ("empty_rows", 1500, 3)])
def test_ghosted_exchange_iterates_like_the_unsharded_matrix(tmp_path, oracle, kind, m, world):
    """Column-footprint exchange (GhostedRowShardedSpmv): three steps of x <- 0.5 A x + 0.25 x with x partitioned like the
    rows, every rank receiving only the x entries its columns reference, equal bit for bit to the unsharded iteration.
    Banded (BASELINE configs[4] family): a rank needs 4 + 3 entries from its neighbours, nothing from anyone else."""
    mp.spawn(_ghost_worker, args=(world, _free_port(), kind, m, str(tmp_path)), nprocs=world, join=True)
    rowptr, cols, vals, x = _ghost_problem(kind, m)
    for _ in range(3):
        x = oracle.host_spmv(0.5, 0.25, rowptr, cols, vals, x, x.copy())
    got = np.empty(m)
    total_bytes = 0
    for r in range(world):
        g = np.load(tmp_path / f"rank{r}.npz")
        got[int(g["r0"]): int(g["r1"])] = g["x"]
        assert int(g["n_ext"]) == int(g["r1"]) - int(g["r0"]) + int(g["n_ghost"])
        total_bytes += int(g["bytes"])
        if kind == "banded":
            inner = 0 < r < world - 1
            assert int(g["n_ghost"]) == (7 if inner else (3 if r == 0 else 4)), (r, int(g["n_ghost"]))
    assert np.array_equal(got, x)
    if kind == "banded":
        assert total_bytes == 8 * 7 * (world - 1)
