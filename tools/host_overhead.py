#!/usr/bin/env python3
"""Host cost of one steady-state call (plan lookup, tunables, launch) on a matrix too small to keep the GPU busy:
20 000 calls without synchronisation, ctypes overhead (a trivial library call) subtracted."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spmv_acc_amd
from spmv_acc_amd import synth
lib = spmv_acc_amd.load_library()
rowptr, cols, vals = synth.random_csr(2000, 2000, 6, seed=3)
nnz = int(rowptr[-1])
drp, dci, dv = (torch.from_numpy(a).cuda() for a in (rowptr, cols, vals))
x = torch.ones(2000, dtype=torch.float64, device="cuda"); y = torch.zeros(2000, dtype=torch.float64, device="cuda")
N = 20000
t0 = time.perf_counter()
for _ in range(N):
    lib.spmv_acc_get_strategy()
base = (time.perf_counter() - t0) / N
for strat in ("adaptive", "line_enhance", "flat", "adaptive_plus", "vector_row"):
    sid = spmv_acc_amd.strategy_id(strat)
    args = (sid, 0, 1.0, 0.0, 2000, 2000, nnz, None, drp.data_ptr(), dci.data_ptr(), dv.data_ptr(), x.data_ptr(), y.data_ptr())
    lib.spmv_acc_csr_spmv_strategy(*args); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        lib.spmv_acc_csr_spmv_strategy(*args)
    t_issue = (time.perf_counter() - t0) / N
    torch.cuda.synchronize()
    t_total = (time.perf_counter() - t0) / N
    print(f"{strat:14s} issue {1e6*(t_issue-base):6.2f} us/call (ctypes baseline {1e6*base:.2f} us)   issue+drain {1e6*t_total:6.2f} us/call")
