#!/usr/bin/env python3
"""Would the whole-row pass of the slab lists (rows < 32 non-zeros, one run each, gathers from all of x) gain from visiting its rows BUCKETED by the
column slab of their last (largest) column -- x locality for the cold gathers without cutting any row?  Emulated with matrices: the short rows of
R-MAT as a matrix of their own, (a) in row order, (b) rows permuted into (bucket, row) order for B = 2 ... 16 buckets; both through the forced slab
passes with every row whole (= the whole-row pass alone), plain and hinted as the plan's timing decides.
    python profiles/probes/whole_rows_bucket_probe.py [scale=25]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
lib = spmv_acc_amd.load_library()
lens = (rp[1:] - rp[:-1]).long()
short = (lens > 0) & (lens < 32)
rows = torch.nonzero(short).flatten()
rp64 = rp.long()
last_col = ci[(rp64[rows + 1] - 1)].long()
x = torch.rand(n, device="cuda", dtype=torch.float64) * 2 - 1


def build(order):
    """CSR of the short rows in the given order (a permutation of `rows`)"""
    L = lens[order]
    nrp = torch.zeros(order.numel() + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(L, 0, out=nrp[1:])
    tot = int(nrp[-1].item())
    src = torch.repeat_interleave(rp64[order] - nrp[:-1], L, output_size=tot) + torch.arange(tot, device="cuda")
    return order.numel(), n, tot, nrp.to(torch.int32), ci[src].contiguous(), v[src].contiguous()


def time_it(A, tag):
    mm, nn, nz, r, c, vv = A
    y0 = torch.zeros(mm, dtype=torch.float64, device="cuda")
    y = y0.clone()
    lib.spmv_acc_reset_tunables()
    lib.spmv_acc_set_tunable(b"slab_segments", 8)
    lib.spmv_acc_set_tunable(b"slab_whole_below", 32)
    spmv_acc_amd.prepare(mm, nn, nz, r, c, vv, x, strategy="line_enhance")
    ms = min(float(np.median(spmv_acc_amd.time_spmv("line_enhance", 8, 1.0, 1.0, mm, nn, nz, r, c, vv, x, y, y0=y0))) for _ in range(2))
    spmv_acc_amd.release_plans(r)
    print(f"{tag:34s} {ms * 1e3:8.1f} us", flush=True)
    return ms


print(f"R-MAT {scale}: {rows.numel()} short rows, {int(lens[rows].sum().item())} non-zeros", flush=True)
time_it(build(rows), "row order (today)")
for B in (2, 4, 8, 16, 32):
    width = (n + B - 1) // B
    bucket = torch.clamp(last_col // width, max=B - 1)
    order = rows[torch.sort(bucket * m + rows).indices]
    time_it(build(order), f"bucketed by last column, B = {B}")
lib.spmv_acc_reset_tunables()
