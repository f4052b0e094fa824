#!/usr/bin/env python3
"""Round 6: R-MAT 25 under line_enhance -- the default path (slab passes over run lists, no copy) against the slab-major COPY
(tunable col_slabs) at several slab counts; per-launch protocol (y reset), kernel clock, back to back.  usage: rmat_col_slabs_sweep.py [scale] [S,S,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
slabs = [int(s) for s in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,8,16".split(","))]
lib = spmv_acc_amd.load_library()
m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
x = torch.rand(n, device="cuda", dtype=torch.float64); y0 = torch.rand(m, device="cuda", dtype=torch.float64); y = y0.clone()
balg = synth.algorithmic_bytes(m, n, nnz)
for S in slabs:
    lib.spmv_acc_reset_tunables()
    if S: assert lib.spmv_acc_set_tunable(b"col_slabs", S) == 0
    torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]
    ms_prep = spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="line_enhance")
    ev, kn, kl = spmv_acc_amd.time_spmv_kernels("line_enhance", 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)
    per = spmv_acc_amd.time_spmv("line_enhance", 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)
    b2b = spmv_acc_amd.time_spmv_total("line_enhance", 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / 8
    free1 = torch.cuda.mem_get_info()[0]
    info = spmv_acc_amd.query_plan(rp, m)
    print(f"R-MAT {scale} col_slabs {S:2d}: pair {np.median(per)*1e3:8.1f} us ({balg/np.median(per)/8e9:.3f})  kernel clock {np.median(kn)*1e3:8.1f} us  b2b {b2b*1e3:8.1f} us  "
          f"launches {int(np.median(kl))}  prepare {ms_prep:.0f} ms  plan memory {(free0-free1)/1e9:.2f} GB  kernel {info['last_kernel']} slab_passes {info['slab_passes']}", flush=True)
    spmv_acc_amd.release_plans(rp); torch.cuda.empty_cache()
