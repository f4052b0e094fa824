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
lib = spmv_acc_amd.load_library()
# (round 3: the opt-in modes too -- run lists, slab-major copies, the full row-pointer digest, LIGHT's counter, a second stream's ordering event)
side = torch.cuda.Stream(); side2 = torch.cuda.Stream()
for it in range(300):
    for s in ("adaptive","flat","adaptive_plus","line_enhance","vector_row","default","light","thread_row"):
        spmv_acc_amd.csr_spmv(1.0,0.0,20000,20000,nnz,drp,dci,dv,x,y,strategy=s)
    for knobs in ({"slab_segments": 4}, {"col_slabs": 3}, {"guard_full": 1}, {"col16": 1}, {"gather_hint": 1}):
        for k, val in knobs.items(): lib.spmv_acc_set_tunable(k.encode(), val)
        spmv_acc_amd.csr_spmv(1.0,0.0,20000,20000,nnz,drp,dci,dv,x,y,strategy="flat" if "col16" in knobs else "line_enhance")
        lib.spmv_acc_reset_tunables()
    # round 4: lazily finished timings (several calls per plan), the two-class run lists, spmv_acc_prepare, the chunks entry on two streams
    for _ in range(4):
        spmv_acc_amd.csr_spmv(1.0,1.0,20000,20000,nnz,drp,dci,dv,x,y,strategy="adaptive")
    spmv_acc_amd.prepare(20000,20000,nnz,drp,dci,dv,x,strategy="flat")
    lib.spmv_acc_set_tunable(b"slab_segments", 8); lib.spmv_acc_set_tunable(b"slab_whole_below", 6)
    spmv_acc_amd.csr_spmv(1.0,1.0,20000,20000,nnz,drp,dci,dv,x,y,strategy="line_enhance")
    lib.spmv_acc_reset_tunables()
    import ctypes
    cuts=[0,7000,7000,20000]; ends=[int(rowptr[c]) for c in cuts[1:]]
    lib.spmv_acc_csr_spmv_chunks(-1,1.0,0.0,20000,3,(ctypes.c_int*4)(*cuts),(ctypes.c_int*3)(*ends),drp.data_ptr(),dci.data_ptr(),dv.data_ptr(),x.data_ptr(),0,y.data_ptr(),
                                 (ctypes.c_void_p*2)(side.cuda_stream, side2.cuda_stream), None)
    for a_,b_ in zip(cuts[:-1],cuts[1:]):
        if b_>a_: spmv_acc_amd.release_plans(drp[a_:])
    lib.spmv_acc_set_stream(side.cuda_stream)
    spmv_acc_amd.csr_spmv(1.0,0.0,20000,20000,nnz,drp,dci,dv,x,y,strategy="flat")
    lib.spmv_acc_set_stream(None)
    # round 5: the kernel clock (its event pool goes with the call), the region timer, the strict / size-rule tunables, an un-rebased view with its own plan
    spmv_acc_amd.time_spmv_kernels("adaptive",3,1.0,1.0,20000,20000,nnz,drp,dci,dv,x,y,y0=x)
    spmv_acc_amd.time_spmv_region("flat",3,1.0,1.0,20000,20000,nnz,drp,dci,dv,x,y)()
    for knobs in ({"strict_strategy": 1}, {"slab_segments": 1, "slab_kb": 8}, {"hint_min_x_mb": 0, "hint_budget_kb": 16}, {"max_grid_blocks": 64, "col_slabs": 2}):
        for k, val in knobs.items(): lib.spmv_acc_set_tunable(k.encode(), val)
        spmv_acc_amd.csr_spmv(1.0,1.0,20000,20000,nnz,drp,dci,dv,x,y,strategy="flat" if "strict_strategy" in knobs else "adaptive_plus")
        lib.spmv_acc_reset_tunables()
    spmv_acc_amd.csr_spmv(1.0,0.0,12000,20000,int(rowptr[20000]),drp[8000:],dci,dv,x,y[8000:],strategy="flat")
    spmv_acc_amd.release_plans(drp[8000:])
    spmv_acc_amd.release_plans(drp)
    if it==20: f0=free()
f1=free()
print("free after 20 cycles:", f0, "after 300:", f1, "delta MB:", (f0-f1)/1e6)
