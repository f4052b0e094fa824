#!/usr/bin/env python3
"""Strategies on a matrix of 70 M rows (more than 2^26: a grid of one wavefront per row would be 2^32 work-items, HIP's limit per launch) with few non-zeros:
every strategy and the slab modes against an independent device evaluation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import spmv_acc_amd
lib = spmv_acc_amd.load_library()
m, n = 70_000_000, 70_000_000
g = torch.Generator(device="cuda"); g.manual_seed(3)
lens = (torch.rand(m, generator=g, device="cuda") < 0.25).long() * torch.randint(1, 6, (m,), generator=g, device="cuda")
lens[12345] = 3000  # (a longer row only slows the checker: its index_add serialises on one address)
rp = torch.zeros(m + 1, dtype=torch.int64, device="cuda"); torch.cumsum(lens, 0, out=rp[1:])
nnz = int(rp[-1].item())
rows = torch.repeat_interleave(torch.arange(m, device="cuda"), lens, output_size=nnz)
ci = torch.randint(0, n, (nnz,), generator=g, device="cuda")
ci = (torch.sort(rows * n + ci).values % n).to(torch.int32)
rp = rp.to(torch.int32)
v = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, generator=g, device="cuda", dtype=torch.float64)
prod = v * x[ci.long()]
ref = y0.clone().index_add_(0, rows, prod)
scale = y0.abs().index_add_(0, rows, prod.abs()) + 1e-300
del prod, rows
print(f"m {m} nnz {nnz}", flush=True)
for strat, knobs in [(s, {}) for s in spmv_acc_amd.STRATEGIES] + [("line_enhance", {"slab_segments": 8}), ("adaptive", {"col_slabs": 4}), ("adaptive", {"guard_full": 1})]:
    lib.spmv_acc_reset_tunables()
    for k, val in knobs.items():
        lib.spmv_acc_set_tunable(k.encode(), val)
    y = y0.clone()
    try:
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        err = float(((y - ref).abs() / scale).max().item())
        info = spmv_acc_amd.query_plan(rp, m) or {}
        print(f"  {strat:20s} {str(knobs):24s} max scaled error {err:.2e}  slab_passes {info.get('slab_passes')}", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"  {strat:20s} {str(knobs):24s} FAILED: {ex}", flush=True)
        lib.spmv_acc_clear_error()
    spmv_acc_amd.release_plans(rp)
lib.spmv_acc_reset_tunables()
