#!/usr/bin/env python3
"""Device-memory leak check: 300 build / release cycles of every kernel family's plan on one matrix; free memory must not move."""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np, spmv_acc_amd
from spmv_acc_amd import synth
rowptr, cols, vals = synth.random_csr(20000, 20000, 9, seed=3, kind="powerlaw")
nnz=int(rowptr[-1])
drp,dci,dv = (torch.from_numpy(a).cuda() for a in (rowptr, cols, vals))
x=torch.ones(20000,dtype=torch.float64,device='cuda'); y=torch.zeros(20000,dtype=torch.float64,device='cuda')
def free(): torch.cuda.synchronize(); return torch.cuda.mem_get_info()[0]
f0=None
for it in range(300):
    for s in ("adaptive","flat","adaptive_plus","line_enhance","vector_row","default"):
        spmv_acc_amd.csr_spmv(1.0,0.0,20000,20000,nnz,drp,dci,dv,x,y,strategy=s)
    spmv_acc_amd.release_plans(drp)
    if it==20: f0=free()
f1=free()
print("free after 20 cycles:", f0, "after 300:", f1, "delta MB:", (f0-f1)/1e6)
