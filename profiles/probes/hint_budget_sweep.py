import sys, numpy as np, torch
sys.path.insert(0, "/root/repo")
import spmv_acc_amd
from spmv_acc_amd import synth
lib = spmv_acc_amd.load_library()
for scale in (25,):
    m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
    x = torch.rand(n, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    for kb in (512, 1024, 2304, 4608, 9216, 18432, 36864, 73728):
        lib.spmv_acc_reset_tunables()
        lib.spmv_acc_set_tunable(b"hint_budget_kb", kb)
        spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="line_enhance")
        ms = min(float(np.median(spmv_acc_amd.time_spmv("line_enhance", 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))) for _ in range(2))
        print(f"R-MAT {scale} hint_budget_kb {kb}: {ms*1e3:.1f} us  slab_passes {spmv_acc_amd.query_plan(rp, m)['slab_passes']}", flush=True)
        spmv_acc_amd.release_plans(rp)
