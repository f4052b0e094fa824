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

    Each rank constructs it with ITS slice.  The gathered vector (world * padded rows) exists twice and the two copies
    alternate: step k computes this rank's rows STRAIGHT INTO its slice of copy k (the out-of-place entry of the C ABI reads the
    old slice from copy k-1, or from ``y_prev``) and the exchange moves that slice to every peer in place -- no staging buffer and
    no device copy anywhere in a step (round 2 copied the slice twice per step: 61 of 218 us on the headline matrix with zero
    bytes exchanged).  While the exchange of step k is in flight step k+1 may already compute into the other copy.

    Within ONE step the exchange can be pipelined against the kernel (``pipeline`` = C > 1): the local rows are cut into C
    chunks, chunk c's kernel runs on the compute stream and its slice travels (point-to-point fan-out, straight to its place in
    every peer's vector) as soon as the kernel of chunk c has finished, i.e. while chunk c+1 computes.  This is the overlap that
    still exists when x_{k+1} depends on the gathered y_k (then nothing of step k+1 can start before the exchange of step k
    has ended; the cross-step overlap above needs an x that does not depend on y -- bench.py's fixed x).
    """

    def __init__(self, rank: int, world: int, bounds, rowptr, cols, vals, n: int, device, strategy="adaptive",
                 local_spmv: Optional[Callable] = None, h_rowptr=None, always_collective: bool = False,
                 exchange: str = "allgather", pipeline: int = 1, own_stream: bool = False):
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
        self.pipeline = max(int(pipeline), 1)
        # two copies of the gathered vector; rows [m_local, pad) of a slice are padding and stay zero (no kernel writes them)
        self._full = [torch.zeros(world * self.pad, dtype=torch.float64, device=device) for _ in range(2)]
        self._pending = None
        self._k = 0
        self._chunks = {}  # pipeline depth -> [(c0, c1, rowptr, cols, vals, nnz, h_rowptr)] over this rank's rows
        # On the GPU the unpipelined step runs the local SpMV on torch's CURRENT stream -- the stream the exchange is issued
        # against: the process group orders the exchange behind it by itself, and `work.wait()` of step k-1's exchange is the only
        # cross-stream dependency a step needs (the two alternating vectors keep SpMV k+1 clear of exchange k).  The engine's
        # OWN (non-NULL) compute stream, with events both ways -- the exchange of step k issued after an event recorded behind
        # SpMV k, SpMV k+2 issued after the stream has waited for that exchange -- serves the pipelined step, where the chunks'
        # kernels must run ahead of the exchanges, and `own_stream=True`.  (Round 2 used it for every step: two extra
        # cross-stream hand-overs per step, ~20 us of bubbles on a 156 us kernel.)
        self._gpu = torch.device(device).type == "cuda"
        self.own_stream = bool(own_stream)
        self.compute_stream = torch.cuda.Stream(device=device) if self._gpu else None
        # the pipelined step alternates its chunks' kernels over two streams: consecutive chunks are independent, and on ONE stream
        # each kernel waits for its predecessor's last wavefront (one rank, headline matrix, C = 8: 0.245 ms against 0.156 for C = 1)
        self.chunk_streams = [self.compute_stream, torch.cuda.Stream(device=device)] if self._gpu else None
        self.exchange_stream = torch.cuda.Stream(device=device) if self._gpu else None
        # where a pipelined step's chunk kernels run: "two" (alternating over the two chunk streams: consecutive chunks overlap their tails; the
        # step pays two cross-stream hand-overs) or "current" (the caller's stream, back to back: no hand-over, but every kernel -> event ->
        # kernel boundary on one stream is a ~13 us bubble).  One rank, headline matrix, dependent step, C = 1 / 2 / 4 / 8:
        # two 0.159 / 0.186 / 0.194 / 0.201 ms, current 0.159 / 0.178 / 0.204 / 0.251 ms (round 3: 0.156 / 0.190 / 0.208 / 0.245)
        self.chunk_stream_mode = "two"
        self.spmv_done = None  # event behind the latest local SpMV (GPU only)
        self.exchange_issued_after_spmv = None  # for tests: did the latest exchange wait for that event?

    # ---- views -------------------------------------------------------------------------------------------------------------
    @property
    def y_full(self):
        """The gathered vector the latest step wrote (complete after ``wait()``)."""
        return self._full[(self._k - 1) & 1] if self._k else self._full[0]

    def _own(self, full):
        return full[self.rank * self.pad: self.rank * self.pad + self.m_local]

    def chunk_bounds(self, depth: int):
        """Row ranges [c0, c1) of the padded slice for a pipeline of ``depth`` chunks -- the same on every rank (the exchange of
        chunk c moves rows [c0, c1) of every rank's slice)."""
        per = -(-self.pad // depth)
        return [(min(c * per, self.pad), min((c + 1) * per, self.pad)) for c in range(depth) if c * per < self.pad]

    def _chunk_arrays(self, depth: int):
        """Per-chunk CSR views of this rank's slice.  A chunk is a row SUB-RANGE of the slice's own arrays: rowptr[a : b + 1] as a
        view (so its first entry is the chunk's first non-zero, not 0), the whole colindex / value arrays, nnz = rowptr[b] -- the
        library's kernels accept that form (tests/test_gpu_parity.py::test_row_shard_without_rebasing), so nothing is copied or
        rebased (round 3 made one rebased rowptr copy per chunk).  Round 5: the library reads the view's first non-zero (rowptr[a]) once per plan
        and sizes every heuristic, flat's tile range and the column census by the chunk's OWN non-zeros (until then `nnz`, the end offset, was read
        as the count: chunk k of C looked k + 1 times as dense as it is)."""
        if depth not in self._chunks:
            out = []
            ends = None
            for c0, c1 in self.chunk_bounds(depth):
                a, b = min(c0, self.m_local), min(c1, self.m_local)
                if depth == 1:
                    out.append((a, b, self.rowptr, self.cols, self.vals, self.nnz_local, self.h_rowptr))
                    continue
                if ends is None:  # every chunk's end offset in ONE device read
                    cuts = sorted({min(c, self.m_local) for bounds in self.chunk_bounds(depth) for c in bounds})
                    vals_at = self.rowptr[cuts] if not hasattr(self.rowptr, "cpu") else self.rowptr[self.torch.as_tensor(cuts, device=self.rowptr.device)].cpu()
                    ends = {c: int(v) for c, v in zip(cuts, vals_at)}
                out.append((a, b, self.rowptr[a: b + 1], self.cols, self.vals, ends[b], None))
            self._chunks[depth] = out
        return self._chunks[depth]

    def prepare(self, beta: float, x, depth: Optional[int] = None) -> float:
        """Build and tune the plan of every chunk of a ``depth``-chunk step (default: the constructor's pipeline) for the beta class of
        ``beta``, up front (spmv_acc_prepare: no tuning budget), so that the steps only enqueue -- plan building allocates, frees and
        synchronises, which must not fall between the exchanges of a step the peers are already in, nor into a timed region.  No
        collective inside.  GPU engine only (a pluggable ``local_spmv`` has no plans).  Returns the device milliseconds spent."""
        if self.local_spmv != self._hip_spmv:
            return 0.0
        depth = self.pipeline if depth is None else max(int(depth), 1)
        ms = 0.0
        for k, (a, b, rp, ci, v, nnz, h_rp) in enumerate(self._chunk_arrays(depth)):
            if b <= a:
                continue
            if self._gpu and depth > 1 and self.chunk_stream_mode == "two":  # on the stream the chunk's kernels will use
                with self.torch.cuda.stream(self.chunk_streams[k & 1]):
                    ms += spmv_acc_amd.prepare(b - a, self.n, nnz, rp, ci, v, x, strategy=self.strategy, h_rowptr=h_rp, beta=beta)
            else:
                ms += spmv_acc_amd.prepare(b - a, self.n, nnz, rp, ci, v, x, strategy=self.strategy, h_rowptr=h_rp, beta=beta)
        return ms

    def _hip_chunks(self, alpha, beta, x, own, y_in, depth, streams):
        """The compute side of a pipelined step in ONE library call (spmv_acc_csr_spmv_chunks): every chunk's kernels on its chunk
        stream, an event behind each.  Per-chunk Python calls cost ~12 us apiece -- 8 chunks of a 160 us step took 245 us with the
        GPU waiting for the host.  Returns the chunks' events (torch events, recorded once at first use so that their handles exist)."""
        import ctypes

        torch = self.torch
        key = ("hip", depth)
        if key not in self._chunks:
            chunks = self._chunk_arrays(depth)
            cuts = [chunks[0][0]] + [c[1] for c in chunks]
            ends = [c[5] for c in chunks]
            evs = []
            for k in range(len(chunks)):
                e = torch.cuda.Event()
                e.record(streams[k & 1])  # (torch creates the underlying hipEvent_t at the first record)
                evs.append(e)
            self._chunks[key] = dict(
                n=len(chunks), cuts=(ctypes.c_int * len(cuts))(*cuts), ends=(ctypes.c_int * len(ends))(*ends),
                events=(ctypes.c_void_p * len(evs))(*[int(e.cuda_event) for e in evs]), torch_events=evs)
        C = self._chunks[key]
        lib = spmv_acc_amd.load_library()
        rc = lib.spmv_acc_csr_spmv_chunks(spmv_acc_amd.strategy_id(self.strategy), alpha, beta, self.n, C["n"], C["cuts"], C["ends"],
                                          self.rowptr.data_ptr(), self.cols.data_ptr(), self.vals.data_ptr(), x.data_ptr(),
                                          0 if y_in is None else y_in.data_ptr(), own.data_ptr(),
                                          (ctypes.c_void_p * 2)(streams[0].cuda_stream, streams[1].cuda_stream), C["events"])
        if rc != 0:
            raise spmv_acc_amd.SpmvAccError(lib.spmv_acc_last_error_string().decode())
        return C["torch_events"]

    def _hip_spmv(self, alpha, beta, x, y_out, y_in, chunk):
        a, b, rp, ci, v, nnz, h_rp = chunk
        spmv_acc_amd.csr_spmv(alpha, beta, b - a, self.n, nnz, rp, ci, v, x, y_out, strategy=self.strategy, h_rowptr=h_rp,
                              y_in=y_in)

    # ---- one step ----------------------------------------------------------------------------------------------------------
    def step(self, alpha: float, beta: float, x, y_prev=None, group=None, overlap: bool = True, pipeline: Optional[int] = None):
        """One sharded SpMV.  ``y_prev`` (m_local values) is this rank's slice of the old y when beta != 0
        (None: iterate in place -- step k reads the y step k-1 produced, starting from the slice given to ``set_y``).
        With ``overlap`` the exchange is left in flight; call ``wait()`` (or the next ``step``) to retire it.
        ``pipeline``: chunks per step (default: the constructor's); > 1 uses the point-to-point fan-out per chunk."""
        depth = self.pipeline if pipeline is None else max(int(pipeline), 1)
        cur, prev = self._full[self._k & 1], self._full[(self._k & 1) ^ 1]
        self._k += 1
        own = self._own(cur)
        y_in = None
        if beta != 0.0:
            y_in = y_prev[: self.m_local] if y_prev is not None else self._own(prev)
        chunks = self._chunk_arrays(depth)
        solo = self.world == 1 and not self.always_collective
        works = []

        def local(chunk):
            a, b = chunk[0], chunk[1]
            if b > a:
                self.local_spmv(alpha, beta, x, own[a:b], None if y_in is None else y_in[a:b], chunk)

        if self._gpu:
            torch = self.torch
            cur_stream = torch.cuda.current_stream(self.device)
            cs = self.compute_stream
            # x / y_prev were produced on the caller's stream; `cur` was last touched by the exchange of step k-2, which the
            # caller's stream has waited for (self.wait() of step k-1)
            if depth == 1 and not self.own_stream:
                local(chunks[0])  # on the current stream (the library follows it, spmv_acc_amd._require)
                self.spmv_done = None
                self.wait()  # at most one exchange in flight (it worked on the OTHER vector: the SpMV above did not wait for it)
                self.exchange_issued_after_spmv = True  # (stream order)
                if not solo:
                    works = self._issue_exchange(cur, group)
            elif depth == 1:
                cs.wait_stream(cur_stream)
                with torch.cuda.stream(cs):  # the library follows torch's current stream (spmv_acc_amd._require)
                    local(chunks[0])
                    self.spmv_done = cs.record_event()
                self.wait()  # at most one exchange in flight
                cur_stream.wait_event(self.spmv_done)  # the exchange (issued against the current stream) starts behind SpMV k
                self.exchange_issued_after_spmv = True
                if not solo:
                    works = self._issue_exchange(cur, group)
            else:
                self.wait()  # (the pipelined exchange starts during this step: the previous one must have ended)
                two = self.chunk_stream_mode == "two"
                if two:
                    for q in self.chunk_streams:
                        q.wait_stream(cur_stream)
                if self.local_spmv == self._hip_spmv:
                    # ONE library call for all chunks; their kernels alternate over the two chunk streams (default) or run back to back on
                    # the current stream (chunk_stream_mode, see __init__)
                    events = self._hip_chunks(alpha, beta, x, own, y_in, depth, (self.chunk_streams if two else [cur_stream, cur_stream]))
                else:
                    events = []
                    for k, ch in enumerate(chunks):  # all kernels go out at once, alternating over the two chunk streams
                        q = self.chunk_streams[k & 1]
                        with torch.cuda.stream(q):
                            local(ch)
                            events.append(q.record_event())
                self.spmv_done = events[-1]
                if two or self.local_spmv != self._hip_spmv:
                    for (c0, c1), ev in zip(self.chunk_bounds(depth), events):
                        cur_stream.wait_event(ev)  # chunk c travels as soon as ITS kernel has finished, while chunk c+1 computes
                        if not solo:
                            works += self._issue_exchange(cur, group, rows=(c0, c1))
                elif not solo:
                    # kernels on the current stream: each chunk's exchange is issued from the exchange stream behind that chunk's event, so it
                    # is ordered after chunk c and not after the chunks enqueued behind it; the caller's stream meets the exchanges in wait()
                    xs = self.exchange_stream
                    with torch.cuda.stream(xs):
                        for (c0, c1), ev in zip(self.chunk_bounds(depth), events):
                            xs.wait_event(ev)
                            works += self._issue_exchange(cur, group, rows=(c0, c1))
                self.exchange_issued_after_spmv = True
        else:
            self.wait()
            for (c0, c1), ch in zip(self.chunk_bounds(depth), chunks):
                local(ch)
                if not solo and depth > 1:
                    works += self._issue_exchange(cur, group, rows=(c0, c1))
            if not solo and depth == 1:
                works = self._issue_exchange(cur, group)
        if solo:
            return None
        self._pending = works
        if not overlap:
            self.wait()
        return works

    def _issue_exchange(self, full, group=None, rows=None):
        """Start moving this rank's slice of ``full`` (or rows [c0, c1) of it) into every peer's copy, in place; returns the
        pending work(s).  A row range always travels point to point: an allgather of a sub-range would land chunk-major."""
        import torch.distributed as dist

        pad, rank, world = self.pad, self.rank, self.world
        if rows is None and self.exchange == "allgather":
            # in place: the input is this rank's slice of the output (the layout RCCL's allgather expects; no self-copy)
            return [dist.all_gather_into_tensor(full, full[rank * pad: (rank + 1) * pad], group=group, async_op=True)]
        c0, c1 = rows if rows is not None else (0, pad)
        if c1 <= c0:
            return []
        ops = []
        for k in range(1, world):  # position k: send to rank+k, receive from rank-k -- a different peer pair per position
            dst, src = (rank + k) % world, (rank - k) % world
            ops.append(dist.P2POp(dist.isend, full[rank * pad + c0: rank * pad + c1], dst, group))
            ops.append(dist.P2POp(dist.irecv, full[src * pad + c0: src * pad + c1], src, group))
        return dist.batch_isend_irecv(ops) if ops else []

    def tune_exchange(self, group=None, warm: int = 2, iters: int = 5):
        """Time both exchange forms on this job's communicator (slice-sized messages, no SpMV), agree on the faster
        across ranks (max over ranks per form) and keep it.  Returns {form: ms per exchange}."""
        import time

        import torch.distributed as dist

        self.wait()
        full = self._full[self._k & 1]  # the copy the next step will write: its content is dead
        on_gpu = full.is_cuda
        result = {}
        for mode in EXCHANGE_MODES:
            self.exchange = mode
            for i in range(warm + iters):
                if i == warm:
                    if on_gpu:
                        self.torch.cuda.synchronize()
                    dist.barrier(group=group)
                    t0 = time.perf_counter()
                for w in self._issue_exchange(full, group):
                    w.wait()
            if on_gpu:
                self.torch.cuda.synchronize()
            t = full.new_tensor([(time.perf_counter() - t0) / iters * 1e3])
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            result[mode] = float(t.item())
        self.exchange = min(result, key=result.get)  # same numbers on every rank -> same choice
        return result

    def tune_pipeline(self, alpha, beta, x, group=None, candidates=(1, 2, 4, 8), warm: int = 2, iters: int = 5):
        """Time whole DEPENDENT steps (kernel + exchange, the next step starting only when the exchange has ended -- the
        x_{k+1} = y_k iteration) for each pipeline depth on this job's communicator, agree across ranks (max over ranks), keep
        the fastest.  Returns {depth: ms per step}.  y is iterated in place from whatever the vectors hold."""
        import time

        import torch.distributed as dist

        result = {}
        for depth in candidates:
            if depth > max(self.pad, 1):
                continue
            self.prepare(beta, x, depth)  # (every chunk's plan settled before the timed steps)
            for i in range(warm + iters):
                if i == warm:
                    if self._gpu:
                        self.torch.cuda.synchronize()
                    if self.world > 1 or self.always_collective:
                        dist.barrier(group=group)
                    t0 = time.perf_counter()
                self.step(alpha, beta, x, group=group, overlap=False, pipeline=depth)
            if self._gpu:
                self.torch.cuda.synchronize()
            t = self._full[0].new_tensor([(time.perf_counter() - t0) / iters * 1e3])
            if self.world > 1 or self.always_collective:
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
            result[depth] = float(t.item())
        # Depth 1 unless a deeper pipeline MEASURES at least 3 % faster (round 6): chunking has a fixed cost -- ~8 us per queue
        # hand-over, 0.159 -> 0.186 / 0.194 / 0.201 ms per step for 2 / 4 / 8 chunks on one rank, DESIGN.md section 5 -- that only an exchange longer
        # than the kernel repays; where the exchange is short (one rank, few ranks on fast links) every depth loses and the choice is 1.
        best = min(result, key=result.get)
        self.pipeline = best if (1 not in result or result[best] < 0.97 * result[1]) else 1
        return result

    def set_y(self, y_slice):
        """Seed this rank's slice of y (for in-place iteration with beta != 0): the next step reads it as its old y."""
        self.wait()
        for full in self._full:  # (both copies: whichever the next step regards as "previous")
            self._own(full).copy_(y_slice[: self.m_local])

    def wait(self):
        if self._pending is not None:
            for w in self._pending:
                w.wait()
            self._pending = None

    def gathered(self):
        """The assembled y (m_global values): padding rows between shards removed."""
        self.wait()
        full = self.y_full
        parts = []
        for r in range(self.world):
            k = int(self.bounds[r + 1] - self.bounds[r])
            parts.append(full[r * self.pad: r * self.pad + k])
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
