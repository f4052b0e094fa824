import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import spmv_acc_amd
from spmv_acc_amd import synth
for scale in (24, 25, 26):
    m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
    x = torch.rand(n, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="line_enhance")
    for rep in range(3):
        ms = float(np.median(spmv_acc_amd.time_spmv("line_enhance", 10, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)))
        print(f"R-MAT {scale} line_enhance {ms*1e3:.1f} us  slab_passes {spmv_acc_amd.query_plan(rp, m)['slab_passes']}", flush=True)
    spmv_acc_amd.release_plans(rp)
    del rp, ci, v, x, y, y0
    torch.cuda.empty_cache()
