#!/usr/bin/env python3
"""Opt-in column-slab blocking (tunable col_slabs) against the default path: R-MAT at a given scale (or a sweep stand-in), per-launch
protocol with y reset, strategies line_enhance / flat / adaptive, S in {0 (off), 4, 8, 12, 16}."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

what = sys.argv[1] if len(sys.argv) > 1 else "25"
if what.isdigit():
    m, n, nnz, rp, ci, v = synth.rmat_torch(int(what), device="cuda", seed=0xC4)
    name = f"R-MAT {what}"
else:
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(what)
    name = what
lib = spmv_acc_amd.load_library()
gen = torch.Generator(device="cuda"); gen.manual_seed(1)
x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, generator=gen, device="cuda", dtype=torch.float64)
b_alg = synth.algorithmic_bytes(m, n, nnz)
print(f"{name}: m {m} n {n} nnz {nnz}", flush=True)
for strat in (sys.argv[2:] or ["line_enhance", "flat"]):
    ref = None
    for S in (0, 4, 8, 12, 16):
        lib.spmv_acc_set_tunable(b"col_slabs", S)
        y = y0.clone()
        for _ in range(3):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        ms = float(np.median(spmv_acc_amd.time_spmv(strat, 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)))
        y.copy_(y0)
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        if ref is None:
            ref = y.clone()
        diff = float(((y - ref).abs() / (ref.abs() + 1.0)).max().item())
        print(f"  {strat:14s} col_slabs {S:2d}: {ms * 1e3:9.1f} us  frac {b_alg / (ms * 1e-3) / 8e12:.4f}  max |diff| / (|y| + 1) vs S = 0: {diff:.1e}", flush=True)
        spmv_acc_amd.release_plans(rp)
lib.spmv_acc_set_tunable(b"col_slabs", 0)
