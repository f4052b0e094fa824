#!/usr/bin/env python3
"""The column-count limit: n = 2,147,483,000 (x = 17 GB), 1 M rows of 8 sorted random columns each -- column arithmetic (slab widths, byte offsets of the
gathers, the census' line index) next to INT_MAX.  Every strategy + the slab modes against an independent device evaluation."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, spmv_acc_amd
lib = spmv_acc_amd.load_library()
m, n, per_row = 1_000_000, 2_147_483_000, 8
g = torch.Generator(device="cuda"); g.manual_seed(5)
nnz = m * per_row
ci = torch.randint(0, n, (m, per_row), generator=g, device="cuda").sort(dim=1).values
ci[:, -1] = n - 1 - torch.arange(m, device="cuda") % 7          # the very last columns too
ci = ci.sort(dim=1).values.reshape(-1).to(torch.int32)
rp = (torch.arange(m + 1, device="cuda") * per_row).to(torch.int32)
v = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, generator=g, device="cuda", dtype=torch.float64)
prod = (v * x[ci.long()]).reshape(m, per_row)
ref = y0 + prod.sum(dim=1)
scale = y0.abs() + prod.abs().sum(dim=1)
print(f"m {m} n {n} nnz {nnz}", flush=True)
for strat, knobs in [(s, {}) for s in spmv_acc_amd.STRATEGIES] + [("line_enhance", {"slab_segments": 16}), ("adaptive", {"col_slabs": 4}), ("flat", {"col16": 1}), ("adaptive", {"gather_hint": 1})]:
    lib.spmv_acc_reset_tunables()
    for k, val in knobs.items():
        lib.spmv_acc_set_tunable(k.encode(), val)
    y = y0.clone()
    try:
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        print(f"  {strat:20s} {str(knobs):26s} max scaled error {float(((y - ref).abs() / scale).max().item()):.2e}", flush=True)
    except Exception as ex:  # noqa: BLE001
        print(f"  {strat:20s} {str(knobs):26s} FAILED: {ex}", flush=True)
        lib.spmv_acc_clear_error()
    spmv_acc_amd.release_plans(rp)
lib.spmv_acc_reset_tunables()
