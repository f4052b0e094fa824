"""Row-range sharded SpMV across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI), one allgather of the y sub-vectors per SpMV.

The reference is single-GPU (hipSetDevice(0) at cli/main.cpp:89; no stream / collective anywhere), so this
is new functionality named by BASELINE.json's north_star.  Rows are independent, so a contiguous row
partition needs no reduction: rank r computes y[r0:r1) from its CSR slice (rebased rowptr, GLOBAL column
ids) and a full copy of x, then every rank receives every slice.  Shards are padded to the same row count
because RCCL has no allgatherv.

The exchange has two forms with the same result (every rank ends with every slice): RCCL's ``all_gather_into_tensor``,
and a direct fan-out -- every rank sends its slice to each of its world-1 peers and receives theirs, all in one
grouped batch of point-to-point operations, staggered so that at every position of the batch each rank talks to a
different peer.  On a fully connected xGMI node (7 links per GPU) the fan-out drives all links at once whatever
ring/tree RCCL would pick for the collective; which one is faster is measured on the job's own communicator
(``tune_exchange``), not assumed.

The local compute is pluggable (``local_spmv``) so the partition / padding / collective logic can be
exercised with the gloo backend on CPU in tests; the default is the HIP library and there is no CPU
fallback in the product path.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

import spmv_acc_amd


EXCHANGE_MODES = ("allgather", "p2p")


def shard_bounds(m: int, world: int, mode: int = 0, h_rowptr=None):
    """Row boundaries of all ranks (world + 1 entries) from the C ABI (spmv_acc_partition_rows):
    mode 0 equal row counts, mode 1 nnz-balanced."""
    return spmv_acc_amd.partition_rows(m, world, mode=mode, h_rowptr=h_rowptr)


def padded_shard_rows(bounds) -> int:
    """Common (padded) shard length so that one equal-count allgather carries every slice."""
    return int(np.max(np.diff(bounds)))


def local_csr_slice(rowptr, cols, vals, r0: int, r1: int):
    """CSR slice of rows [r0, r1): rowptr rebased to start at 0, global column ids kept.
    Works on numpy arrays and torch tensors alike."""
    s, e = int(rowptr[r0]), int(rowptr[r1])
    rp = rowptr[r0: r1 + 1] - rowptr[r0]
    return rp, cols[s:e], vals[s:e]


class RowShardedSpmv:
    """y_full = alpha * A * x + beta * y_full, A row-sharded over the ranks of ``group``.

    Each rank constructs it with ITS slice.  ``y_local`` (padded) and ``y_full`` (world * padded) are
    torch tensors on the rank's device; ``step`` enqueues the local SpMV and the allgather.
    """

    def __init__(self, rank: int, world: int, bounds, rowptr, cols, vals, n: int, device, strategy="adaptive",
                 local_spmv: Optional[Callable] = None, h_rowptr=None, always_collective: bool = False,
                 exchange: str = "allgather"):
        import torch

        self.torch = torch
        self.rank, self.world = rank, world
        self.bounds = np.asarray(bounds)
        self.r0, self.r1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.m_local = self.r1 - self.r0
        self.m_global = int(self.bounds[-1])
        self.pad = padded_shard_rows(self.bounds)
        self.n = n
        self.rowptr, self.cols, self.vals = rowptr, cols, vals
        self.nnz_local = int(rowptr[self.m_local])
        self.h_rowptr = h_rowptr
        self.strategy = strategy
        self.device = device
        self.local_spmv = local_spmv or self._hip_spmv
        self.always_collective = always_collective  # issue the allgather even on a one-rank group (RCCL rehearsal)
        if exchange not in EXCHANGE_MODES:
            raise ValueError(f"exchange must be one of {EXCHANGE_MODES}")
        self.exchange = exchange
        # two y buffers: the allgather of step k may still be reading one while step k+1 writes the other
        self.y_local = [torch.zeros(self.pad, dtype=torch.float64, device=device) for _ in range(2)]
        self.y_full = torch.zeros(world * self.pad, dtype=torch.float64, device=device)
        self._pending = None
        self._k = 0
        # On the GPU the local SpMV runs on its OWN (non-NULL) stream and is ordered against the exchange with events, both
        # ways: the exchange of step k is issued after an event recorded behind SpMV k, and SpMV k+2 -- the next writer of
        # the buffer that exchange reads -- is issued after the stream has waited for that exchange.  Nothing relies on the
        # library's stream and torch's current stream being the same stream.
        self._gpu = torch.device(device).type == "cuda"
        self.compute_stream = torch.cuda.Stream(device=device) if self._gpu else None
        self.spmv_done = None  # event behind the latest local SpMV (GPU only)
        self.exchange_issued_after_spmv = None  # for tests: did the latest exchange wait for that event?

    def _hip_spmv(self, alpha, beta, x, y):
        spmv_acc_amd.csr_spmv(alpha, beta, self.m_local, self.n, self.nnz_local, self.rowptr, self.cols, self.vals, x, y,
                              strategy=self.strategy, h_rowptr=self.h_rowptr)

    def step(self, alpha: float, beta: float, x, y_prev=None, group=None, overlap: bool = True):
        """One sharded SpMV.  ``y_prev`` (m_local values) is this rank's slice of the old y when beta != 0
        (None: iterate in place -- step k reads the y step k-1 produced, starting from the slice given to ``set_y``).
        With ``overlap`` the allgather is left in flight; call ``wait()`` (or the next ``step``) to retire it."""
        import torch.distributed as dist

        buf = self.y_local[self._k & 1]
        prev = self.y_local[(self._k & 1) ^ 1]
        first = self._k == 0
        self._k += 1

        def local():
            if beta != 0.0:
                if y_prev is not None:
                    buf[: self.m_local].copy_(y_prev[: self.m_local])
                elif not first:
                    # in-place iteration: the old y of step k is the RESULT of step k-1, which lives in the other buffer
                    # (the two buffers alternate so that the exchange of k-1 can still be reading it: a read, like this copy)
                    buf[: self.m_local].copy_(prev[: self.m_local])
                # (first step: both buffers hold the slice set_y seeded)
            if self.m_local > 0:
                self.local_spmv(alpha, beta, x, buf)

        if self._gpu:
            torch = self.torch
            cur = torch.cuda.current_stream(self.device)
            cs = self.compute_stream
            # x / y_prev were produced on the caller's stream; `buf` was last read by the exchange of step k-2, which the
            # caller's stream has waited for (self.wait() of step k-1)
            cs.wait_stream(cur)
            with torch.cuda.stream(cs):  # the library follows torch's current stream (spmv_acc_amd._require)
                local()
                self.spmv_done = cs.record_event()
            self.wait()  # at most one exchange in flight: y_full is written by it
            cur.wait_event(self.spmv_done)  # the exchange (issued against the current stream) starts behind SpMV k
            self.exchange_issued_after_spmv = True
        else:
            local()
            self.wait()
        if self.world == 1 and not self.always_collective:
            self.y_full[: self.pad].copy_(buf)
            return None
        work = self._issue_exchange(buf, group)
        self._pending = work
        if not overlap:
            self.wait()
        return work

    def _issue_exchange(self, buf, group=None):
        """Start moving ``buf`` (this rank's padded slice) into every rank's y_full; returns the pending work(s)."""
        import torch.distributed as dist

        if self.exchange == "allgather":
            return [dist.all_gather_into_tensor(self.y_full, buf, group=group, async_op=True)]
        pad, rank, world = self.pad, self.rank, self.world
        self.y_full[rank * pad: (rank + 1) * pad].copy_(buf)
        ops = []
        for k in range(1, world):  # position k: send to rank+k, receive from rank-k -- a different peer pair per position
            dst, src = (rank + k) % world, (rank - k) % world
            ops.append(dist.P2POp(dist.isend, buf, dst, group))
            ops.append(dist.P2POp(dist.irecv, self.y_full[src * pad: (src + 1) * pad], src, group))
        return dist.batch_isend_irecv(ops) if ops else []

    def tune_exchange(self, group=None, warm: int = 2, iters: int = 5):
        """Time both exchange forms on this job's communicator (slice-sized messages, no SpMV), agree on the faster
        across ranks (max over ranks per form) and keep it.  Returns {form: ms per exchange}."""
        import time

        import torch.distributed as dist

        self.wait()
        buf = self.y_local[0]
        on_gpu = buf.is_cuda
        result = {}
        for mode in EXCHANGE_MODES:
            self.exchange = mode
            for i in range(warm + iters):
                if i == warm:
                    if on_gpu:
                        self.torch.cuda.synchronize()
                    dist.barrier(group=group)
                    t0 = time.perf_counter()
                for w in self._issue_exchange(buf, group):
                    w.wait()
            if on_gpu:
                self.torch.cuda.synchronize()
            t = self.y_full.new_tensor([(time.perf_counter() - t0) / iters * 1e3])
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            result[mode] = float(t.item())
        self.exchange = min(result, key=result.get)  # same numbers on every rank -> same choice
        return result

    def set_y(self, y_slice):
        """Seed both y buffers with this rank's slice of y (for in-place iteration with beta != 0)."""
        for b in self.y_local:
            b[: self.m_local].copy_(y_slice[: self.m_local])

    def wait(self):
        if self._pending is not None:
            for w in self._pending:
                w.wait()
            self._pending = None

    def gathered(self):
        """The assembled y (m_global values): padding rows between shards removed."""
        self.wait()
        parts = []
        for r in range(self.world):
            k = int(self.bounds[r + 1] - self.bounds[r])
            parts.append(self.y_full[r * self.pad: r * self.pad + k])
        return self.torch.cat(parts)


class GhostedRowShardedSpmv:
    """Row-sharded SpMV for SQUARE matrices whose x is partitioned like the rows (x_{k+1} = f(y_k) solvers): instead of
    every rank receiving every y slice, each rank receives only the entries of x its columns reference.

    Plan (once per matrix): the distinct column ids outside this rank's own range ("ghosts", sorted, hence grouped by
    owner) are found on the device; ranks exchange how many and which entries they need from each other; colindex is
    rewritten once into local numbering -- own columns to [0, n_local), ghosts to n_local + their rank in the sorted ghost
    list -- so the local SpMV is the ordinary library call on an (m_local) x (n_local + n_ghost) matrix.
    Step: pack the entries the peers asked for (one gather), one grouped batch of point-to-point sends / receives straight
    into the ghost segment of x_ext (the ghosts of one owner are contiguous there), local SpMV.
    For the banded matrix of BASELINE configs[4] the exchange is 4 + 3 doubles per neighbour instead of 256 MB per peer.

    Same summation order per row as the unsharded matrix, so results are bit-identical to it.  Arrays are torch
    tensors (CPU with gloo in tests, CUDA with RCCL in use); ``local_spmv`` is pluggable as in ``RowShardedSpmv``.
    """

    def __init__(self, rank: int, world: int, bounds, rowptr, cols, vals, device, strategy="adaptive",
                 local_spmv: Optional[Callable] = None, group=None):
        import torch
        import torch.distributed as dist

        self.torch = torch
        self.rank, self.world, self.group = rank, world, group
        self.bounds = np.asarray(bounds, dtype=np.int64)
        self.c0, self.c1 = int(self.bounds[rank]), int(self.bounds[rank + 1])
        self.m_local = self.n_local = self.c1 - self.c0
        self.rowptr, self.vals = rowptr, vals
        self.nnz_local = int(rowptr[self.m_local])
        self.strategy, self.device = strategy, device
        self.local_spmv = local_spmv or self._hip_spmv

        cols64 = cols.to(torch.int64)
        remote = (cols64 < self.c0) | (cols64 >= self.c1)
        ghosts = torch.unique(cols64[remote])  # sorted global ids
        self.n_ghost = int(ghosts.numel())
        tb = torch.as_tensor(self.bounds, device=ghosts.device)
        cut = torch.searchsorted(ghosts, tb)  # ghosts owned by p: [cut[p], cut[p+1])
        want = (cut[1:] - cut[:-1]).to(torch.int64)  # entries this rank needs from each owner (0 for itself)
        # everyone learns the whole want matrix: asked[q] = what rank q needs from me
        table = [torch.zeros_like(want) for _ in range(world)]
        if world > 1:
            dist.all_gather(table, want, group=group)
        else:
            table[0] = want
        self.recv_counts = [int(v) for v in want.tolist()]
        self.send_counts = [int(table[q][rank].item()) for q in range(world)]
        self.ghost_off = [int(v) for v in cut.tolist()]
        # tell every owner WHICH of its entries this rank needs (owner-local indices), learn what the others need from me
        send_idx = [torch.empty(self.send_counts[q], dtype=torch.int64, device=ghosts.device) for q in range(world)]
        ops, keep = [], []
        for k in range(1, world):
            dst, src = (rank + k) % world, (rank - k) % world
            if self.recv_counts[dst] > 0:
                need = (ghosts[self.ghost_off[dst]: self.ghost_off[dst + 1]] - int(self.bounds[dst])).contiguous()
                keep.append(need)
                ops.append(dist.P2POp(dist.isend, need, dst, group))
            if self.send_counts[src] > 0:
                ops.append(dist.P2POp(dist.irecv, send_idx[src], src, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        self.send_idx = torch.cat(send_idx) if world > 1 else send_idx[0]  # grouped by destination rank
        self.send_off = np.concatenate([[0], np.cumsum(self.send_counts)]).astype(np.int64)
        # colindex in local numbering
        local = torch.where(remote, self.n_local + torch.searchsorted(ghosts, cols64), cols64 - self.c0)
        self.cols_local = local.to(torch.int32).contiguous()
        self.x_ext = torch.zeros(self.n_local + self.n_ghost, dtype=torch.float64, device=device)
        self._x_other = None  # second buffer, allocated by iterate()
        self.send_buf = torch.empty(int(self.send_off[-1]), dtype=torch.float64, device=device)
        self.exchanged_bytes_per_step = 8 * int(self.send_off[-1])

    def _hip_spmv(self, alpha, beta, x, y):
        spmv_acc_amd.csr_spmv(alpha, beta, self.m_local, self.n_local + self.n_ghost, self.nnz_local, self.rowptr,
                              self.cols_local, self.vals, x, y, strategy=self.strategy)

    def set_x(self, x_local):
        """This rank's slice of x (n_local values)."""
        self.x_ext[: self.n_local].copy_(x_local[: self.n_local])

    def exchange(self):
        """Bring the ghost entries of x_ext up to date with the owners' current slices."""
        import torch.distributed as dist

        if self.world == 1:
            return
        torch = self.torch
        if self.send_buf.numel():
            torch.index_select(self.x_ext[: self.n_local], 0, self.send_idx, out=self.send_buf)
        ops = []
        for k in range(1, self.world):
            dst, src = (self.rank + k) % self.world, (self.rank - k) % self.world
            if self.send_counts[dst] > 0:
                ops.append(dist.P2POp(dist.isend, self.send_buf[int(self.send_off[dst]): int(self.send_off[dst + 1])], dst, self.group))
            if self.recv_counts[src] > 0:
                seg = self.x_ext[self.n_local + self.ghost_off[src]: self.n_local + self.ghost_off[src + 1]]
                ops.append(dist.P2POp(dist.irecv, seg, src, self.group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def iterate(self, alpha: float):
        """x <- alpha * A * x in place across the ranks (power-iteration shape, beta = 0): the product is written straight
        into the own-slice of a second x buffer, which then becomes the current one -- no copy between steps."""
        if self._x_other is None:
            self._x_other = self.torch.zeros_like(self.x_ext)
        self.exchange()
        if self.m_local > 0:
            self.local_spmv(alpha, 0.0, self.x_ext, self._x_other[: self.n_local])
        self.x_ext, self._x_other = self._x_other, self.x_ext
        return self.x_ext[: self.n_local]

    def step(self, alpha: float, beta: float, y_local):
        """y_local = alpha * A_local * x + beta * y_local with x = the owners' current slices (set_x on every rank)."""
        self.exchange()
        if self.m_local > 0:
            self.local_spmv(alpha, beta, self.x_ext, y_local)
        return y_local
