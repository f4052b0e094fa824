import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, spmv_acc_amd
from spmv_acc_amd import synth
A = synth.sweep_standin_torch("Hardesty3")
m, n, nnz, rp, ci, v = A
x = torch.rand(n, device="cuda", dtype=torch.float64); y = torch.zeros(m, device="cuda", dtype=torch.float64)
for strat in ("flat", "adaptive"):
    spmv_acc_amd.release_plans(rp)
    for i in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        print(f"=== {strat} call {i+1}: {(time.perf_counter()-t0)*1e3:.3f} ms", file=sys.stderr, flush=True)
