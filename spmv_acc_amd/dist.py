"""Row-range sharded SpMV across the GPUs of one node: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI), one allgather of the y sub-vectors per SpMV.

The reference is single-GPU (hipSetDevice(0) at cli/main.cpp:89; no stream / collective anywhere), so this
is new functionality named by BASELINE.json's north_star.  Rows are independent, so a contiguous row
partition needs no reduction: rank r computes y[r0:r1) from its CSR slice (rebased rowptr, GLOBAL column
ids) and a full copy of x, then every rank receives every slice.  Shards are padded to the same row count
because RCCL has no allgatherv.

The local compute is pluggable (``local_spmv``) so the partition / padding / collective logic can be
exercised with the gloo backend on CPU in tests; the default is the HIP library and there is no CPU
fallback in the product path.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np

import spmv_acc_amd


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
                 local_spmv: Optional[Callable] = None, h_rowptr=None, always_collective: bool = False):
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
        # two y buffers: the allgather of step k may still be reading one while step k+1 writes the other
        self.y_local = [torch.zeros(self.pad, dtype=torch.float64, device=device) for _ in range(2)]
        self.y_full = torch.zeros(world * self.pad, dtype=torch.float64, device=device)
        self._pending = None
        self._k = 0

    def _hip_spmv(self, alpha, beta, x, y):
        spmv_acc_amd.csr_spmv(alpha, beta, self.m_local, self.n, self.nnz_local, self.rowptr, self.cols, self.vals, x, y,
                              strategy=self.strategy, h_rowptr=self.h_rowptr)

    def step(self, alpha: float, beta: float, x, y_prev=None, group=None, overlap: bool = True):
        """One sharded SpMV.  ``y_prev`` (m_local values) is this rank's slice of the old y when beta != 0
        (None: iterate in place on the step's y buffer, see ``set_y``).
        With ``overlap`` the allgather is left in flight; call ``wait()`` (or the next ``step``) to retire it."""
        import torch.distributed as dist

        buf = self.y_local[self._k & 1]
        self._k += 1
        if beta != 0.0 and y_prev is not None:
            buf[: self.m_local].copy_(y_prev[: self.m_local])
        # beta != 0 with y_prev None: the buffer's current content is the old y slice (in-place iteration)
        if self.m_local > 0:
            self.local_spmv(alpha, beta, x, buf)
        self.wait()  # at most one allgather in flight: y_full is written by it
        if self.world == 1 and not self.always_collective:
            self.y_full[: self.pad].copy_(buf)
            return None
        work = dist.all_gather_into_tensor(self.y_full, buf, group=group, async_op=True)
        self._pending = work
        if not overlap:
            self.wait()
        return work

    def set_y(self, y_slice):
        """Seed both y buffers with this rank's slice of y (for in-place iteration with beta != 0)."""
        for b in self.y_local:
            b[: self.m_local].copy_(y_slice[: self.m_local])

    def wait(self):
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    def gathered(self):
        """The assembled y (m_global values): padding rows between shards removed."""
        self.wait()
        parts = []
        for r in range(self.world):
            k = int(self.bounds[r + 1] - self.bounds[r])
            parts.append(self.y_full[r * self.pad: r * self.pad + k])
        return self.torch.cat(parts)
