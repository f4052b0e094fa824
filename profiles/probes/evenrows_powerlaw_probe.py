#!/usr/bin/env python3
"""Matrices with EVEN row lengths and power-law columns (x beyond the caches): does the column-slab pass over run lists pay where the
rows need no rescue?  line_enhance / adaptive / adaptive_plus as shipped against slab_segments forced."""
import sys, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import spmv_acc_amd
lib = spmv_acc_amd.load_library()
for (m, n, per_row, power) in ((300_000, 16_000_000, 40, 6), (2_000_000, 33_000_000, 16, 6), (2_000_000, 33_000_000, 16, 3), (4_000_000, 16_000_000, 8, 12)):
    g = torch.Generator(device="cuda"); g.manual_seed(11)
    nnz = m * per_row
    rows = torch.arange(m, device="cuda").repeat_interleave(per_row)
    ci = (torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) ** power * n).long().clamp_(0, n - 1)
    ci = (torch.sort(rows * n + ci).values % n).to(torch.int32)
    rp = (torch.arange(m + 1, device="cuda") * per_row).to(torch.int32)
    v = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, generator=g, device="cuda", dtype=torch.float64)
    out = []
    for strat, knobs in (("line_enhance", {}), ("adaptive", {}), ("adaptive_plus", {}), ("line_enhance", {"slab_segments": 4}), ("line_enhance", {"slab_segments": 8})):
        lib.spmv_acc_reset_tunables()
        for k, val in knobs.items():
            lib.spmv_acc_set_tunable(k.encode(), val)
        y = y0.clone()
        for _ in range(3):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        ms = float(np.median(spmv_acc_amd.time_spmv(strat, 20, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)))
        info = spmv_acc_amd.query_plan(rp, m)
        out.append(f"{strat}{knobs or ''} {ms*1e3:.1f} us (slab_passes {info['slab_passes']})")
        spmv_acc_amd.release_plans(rp)
    lib.spmv_acc_reset_tunables()
    print(f"m {m} n {n} {per_row}/row power {power}: " + "; ".join(out), flush=True)
