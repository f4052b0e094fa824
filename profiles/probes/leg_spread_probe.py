#!/usr/bin/env python3
"""How much of a bench leg's figure is WHERE its vectors and plan tables happen to lie: one stand-in, one strategy, N fresh (y, plan) pairs in one process --
(a) y re-cloned and the plan rebuilt each time, as bench.py's legs do; (b) one y kept, plan rebuilt; (c) y re-cloned, plan kept.  Per-launch protocol, median of 60."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, spmv_acc_amd
from spmv_acc_amd import synth
name = sys.argv[1] if len(sys.argv) > 1 else "Bump_2911"
strat = sys.argv[2] if len(sys.argv) > 2 else "adaptive"
m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
def leg(y, fresh_plan):
    if fresh_plan:
        spmv_acc_amd.release_plans(rp)
    for _ in range(10):
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    return float(np.median(spmv_acc_amd.time_spmv(strat, 60, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))) * 1e3
keep = []
a = []
for i in range(8):
    y = y0.clone(); keep.append(torch.empty(1 + 3_000_017 * (i + 1), dtype=torch.uint8, device="cuda"))  # (shift what the next clone gets)
    a.append(leg(y, True))
y = y0.clone()
b = [leg(y, True) for _ in range(8)]
c = []
for i in range(8):
    y = y0.clone(); keep.append(torch.empty(1 + 5_000_011 * (i + 1), dtype=torch.uint8, device="cuda"))
    c.append(leg(y, False))
fmt = lambda t: " ".join(f"{q:.1f}" for q in t) + f"   spread {100 * (max(t) / min(t) - 1):.1f} %"
print(f"{name} {strat}: (a) fresh y + fresh plan: {fmt(a)}")
print(f"{name} {strat}: (b) same y, fresh plan:  {fmt(b)}")
print(f"{name} {strat}: (c) fresh y, same plan:  {fmt(c)}")
