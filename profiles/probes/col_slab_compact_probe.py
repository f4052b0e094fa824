#!/usr/bin/env python3
"""Would COMPACT column slabs (only the rows that have non-zeros in the slab, with a row-id list) pay at larger slab counts?
Lower-bound probe: per slab, the shipped kernels run the compacted sub-matrix into a compact y_s (beta = 0); the merge of y_s into y
through the row ids is costed at its bytes (8 B y_s + 4 B row id + 16 B y read-modify-write per non-empty row) over 6 TB/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
strat = "line_enhance"
m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
x = torch.rand(n, device="cuda", dtype=torch.float64)
rows = torch.repeat_interleave(torch.arange(m, device="cuda", dtype=torch.int32), (rp[1:] - rp[:-1]).to(torch.int64))
print(f"R-MAT {scale}: m {m} nnz {nnz}", flush=True)
for S in (8, 16, 32, 64):
    width = -(-n // S)
    slab = (ci // width).to(torch.int32)
    total_us, pairs = 0.0, 0
    for s in range(S):
        sel = slab == s
        r = rows[sel].to(torch.int64)
        if r.numel() == 0:
            continue
        ids, counts = torch.unique_consecutive(r, return_counts=True)
        ms_ = int(ids.numel())
        rps = torch.zeros(ms_ + 1, dtype=torch.int32, device="cuda")
        rps[1:] = torch.cumsum(counts, 0).to(torch.int32)
        cis, vs = ci[sel].contiguous(), v[sel].contiguous()
        ys = torch.zeros(ms_, dtype=torch.float64, device="cuda")
        nz = int(rps[-1].item())
        for _ in range(3):
            spmv_acc_amd.csr_spmv(1.0, 0.0, ms_, n, nz, rps, cis, vs, x, ys, strategy=strat)
        torch.cuda.synchronize()
        t = float(np.median(spmv_acc_amd.time_spmv(strat, 6, 1.0, 0.0, ms_, n, nz, rps, cis, vs, x, ys)))
        total_us += t * 1e3
        pairs += ms_
        spmv_acc_amd.release_plans(rps)
        del sel, r, ids, counts, rps, cis, vs, ys
    merge_us = pairs * 28 / 6e12 * 1e6
    print(f"  {S:3d} compact slabs: kernels {total_us:8.1f} us + merge >= {merge_us:7.1f} us ({pairs / 1e6:.0f} M non-empty (row, slab) pairs) = {total_us + merge_us:8.1f} us", flush=True)
    torch.cuda.empty_cache()
