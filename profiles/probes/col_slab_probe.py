#!/usr/bin/env python3
"""Feasibility probe for column-slab blocking on power-law columns (R-MAT): A = sum_s A_s with A_s the non-zeros whose x line
falls in slab s ((line >> 0) & (S-1) interleaved, or contiguous column ranges), y = beta*y + alpha * sum_s A_s x run as S
sequential SpMVs of the shipped kernels.  Every phase gathers from 1/S of x, so each L2 holds a hot set S times deeper.
Sub-matrices are built here with torch (a library version would build them once per plan)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
strat = sys.argv[2] if len(sys.argv) > 2 else "line_enhance"
m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, generator=gen, device="cuda", dtype=torch.float64)
y = y0.clone()
for _ in range(3):
    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
torch.cuda.synchronize()
base = np.median(spmv_acc_amd.time_spmv(strat, 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))
ref = y0.clone()
spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, ref, strategy=strat)
torch.cuda.synchronize()
print(f"R-MAT {scale}: m {m} nnz {nnz}  {strat} whole matrix: {base * 1e3:.1f} us", flush=True)
rows = torch.repeat_interleave(torch.arange(m, device="cuda", dtype=torch.int32), (rp[1:] - rp[:-1]).to(torch.int64))
for S, mode in ((8, "interleaved"), (8, "contiguous"), (4, "interleaved"), (16, "interleaved")):
    if mode == "interleaved":
        slab = (ci >> 4) & (S - 1)
    else:
        width = -(-n // S)
        slab = (ci // width).to(torch.int32)
    subs = []
    for s in range(S):
        sel = slab == s
        r = rows[sel]
        cnt = torch.bincount(r.to(torch.int64), minlength=m)
        rps = torch.zeros(m + 1, dtype=torch.int32, device="cuda")
        rps[1:] = torch.cumsum(cnt, 0).to(torch.int32)
        subs.append((int(rps[-1].item()), rps, ci[sel].contiguous(), v[sel].contiguous()))
        del sel, r, cnt
    torch.cuda.synchronize()

    def run(yy):
        for k, (nz, rps, cis, vs) in enumerate(subs):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nz, rps, cis, vs, x, yy, strategy=strat)

    yy = y0.clone()
    for _ in range(3):
        run(yy)
    torch.cuda.synchronize()
    yy = y0.clone()
    run(yy)
    torch.cuda.synchronize()
    err = float(((yy - ref).abs() / (ref.abs() + 1e-300)).max().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(6):
        yy.copy_(y0)
        e0.record()
        run(yy)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"  {S:2d} slabs {mode:11s}: {np.median(ts) * 1e3:8.1f} us  ({base / np.median(ts):.2f}x)  max rel diff vs whole-matrix result {err:.2e}  nnz per slab {[s_[0] for s_ in subs]}", flush=True)
    for s_ in subs:
        spmv_acc_amd.release_plans(s_[1])
    del subs
    torch.cuda.empty_cache()
