#!/usr/bin/env python3
"""Balanced rows, power-law columns (every row 16 non-zeros, column bits 1 with probability 0.24: R-MAT's column marginal without its row skew):
which strategies reach the gather hints?   python profiles/probes/colskew_bench.py [log2 n = 25]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spmv_acc_amd

bits = int(sys.argv[1]) if len(sys.argv) > 1 else 25
n = 1 << bits
m = n // 2
per_row = 16
nnz = m * per_row
g = torch.Generator(device="cuda"); g.manual_seed(5)
ci = torch.zeros(nnz, dtype=torch.int32, device="cuda")
for b in range(bits):
    ci |= (torch.rand(nnz, generator=g, device="cuda") < 0.24).to(torch.int32) << b
ci = ci.view(m, per_row).sort(dim=1).values.reshape(-1).contiguous()
rp = (torch.arange(m + 1, device="cuda", dtype=torch.int64) * per_row).to(torch.int32)
v = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64)
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64)
y = torch.zeros(m, device="cuda", dtype=torch.float64)
lib = spmv_acc_amd.load_library()
ref = None
for strat in ("line_enhance", "adaptive", "adaptive_plus", "flat"):
    for hint in (0, -1):
        lib.spmv_acc_reset_tunables()
        assert lib.spmv_acc_set_tunable(b"gather_hint", hint) == 0
        spmv_acc_amd.release_plans(rp)
        for _ in range(3):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        t = spmv_acc_amd.time_spmv_total(strat, 20, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / 20
        y.zero_(); spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, nnz, rp, ci, v, x, y, strategy=strat); torch.cuda.synchronize()
        if ref is None: ref = y.clone()
        err = float((y - ref).abs().max() / ref.abs().max())
        print(f"{strat:14s} gather_hint={hint:2d}  {t * 1e3:9.1f} us   max rel diff to first {err:.1e}", flush=True)
        y.zero_()
lib.spmv_acc_reset_tunables()
