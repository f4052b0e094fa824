#!/usr/bin/env python3
"""Column-slab blocking without a copy (tunable slab_segments, k_segment.hip) against the default path and the slab-major copy
(col_slabs): R-MAT at a given scale or a sweep stand-in, per-launch protocol with y reset.
    python tools/slab_segment_bench.py [scale | stand-in] [strategy]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

what = sys.argv[1] if len(sys.argv) > 1 else "25"
strat = sys.argv[2] if len(sys.argv) > 2 else "line_enhance"
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
print(f"{name}: m {m} n {n} nnz {nnz}  strategy {strat}", flush=True)
variants = [{}] + [{"slab_segments": S} for S in (4, 8, 12, 16)] + [{"col_slabs": 8}]
if len(sys.argv) > 3:
    variants = [dict(kv.split("=") for kv in var.split(",") if kv) for var in sys.argv[3].split(";")]
    variants = [{k: int(v_) for k, v_ in var.items()} for var in variants]
ref = None
for var in variants:
    lib.spmv_acc_reset_tunables()
    for k, val in var.items():
        assert lib.spmv_acc_set_tunable(k.encode(), val) == 0, k
    y = y0.clone()
    t_first = spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
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
    print(f"  {str(var or 'default'):48s}: {ms * 1e3:9.1f} us  frac {b_alg / (ms * 1e-3) / 8e12:.4f}  first call {t_first:8.1f} ms  max |diff| / (|y| + 1) vs default: {diff:.1e}", flush=True)
    spmv_acc_amd.release_plans(rp)
lib.spmv_acc_reset_tunables()
