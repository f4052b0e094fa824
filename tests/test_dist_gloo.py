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


def _worker(rank, world, port, mode, m, out_path, exchange="allgather"):
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

    def local_spmv(alpha, beta, xt, yt):  # oracle stands in for the HIP kernel on CPU
        yl = yt.numpy()[: r1 - r0]
        oracle_lib.host_spmv_inplace(alpha, beta, rp, ci, v, xt.numpy(), yl)

    eng = RowShardedSpmv(rank, world, bounds, rp, ci, v, n, torch.device("cpu"), local_spmv=local_spmv,
                         exchange="allgather" if exchange == "tune" else exchange)
    tuned = eng.tune_exchange(warm=1, iters=2) if exchange == "tune" else None
    xt = torch.from_numpy(x)
    ylocal = torch.from_numpy(y0[r0:r1].copy())
    results = []
    for alpha, beta in ((1.0, 1.0), (0.5, 0.0), (2.0, -1.0)):
        eng.step(alpha, beta, xt, y_prev=ylocal, overlap=True)
        eng.step(alpha, beta, xt, y_prev=ylocal, overlap=True)  # second step exercises the double buffer
        results.append(eng.gathered().numpy().copy())
    if rank == 0:
        np.savez(out_path, bounds=bounds, exchange=np.array(eng.exchange), tuned=np.array(sorted(tuned) if tuned else []),
                 **{f"y{i}": r for i, r in enumerate(results)})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,m,world,exchange", [(0, 4001, 2, "allgather"), (1, 4001, 2, "allgather"), (0, 4096, 2, "allgather"),
                                                   (0, 4001, 2, "p2p"), (1, 4001, 3, "p2p"), (0, 4099, 3, "tune")])
def test_row_sharded_spmv_world2(tmp_path, oracle, mode, m, world, exchange):
    """world 2 and 3, both exchange forms (RCCL allgather / direct fan-out) and the timed choice between them."""
    from spmv_acc_amd import synth

    out = str(tmp_path / "y.npz")
    mp.spawn(_worker, args=(world, _free_port(), mode, m, out, exchange), nprocs=world, join=True)
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
