#!/usr/bin/env python3
"""Does WHERE the matrix lives matter?  The same stand-in generated several times in one process behind pads of different sizes (so its
arrays land at different virtual -- and physical -- addresses), the same pinned plan each time; back-to-back and per-launch times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "Bump_2911"
strat = sys.argv[2] if len(sys.argv) > 2 else "line_enhance"
lib = spmv_acc_amd.load_library()
lib.spmv_acc_set_tunable(b"deterministic", 1)
keep = []
for k, padmb in enumerate((0, 1, 5, 64, 200, 777, 1500, 3000, 0, 0)):
    if padmb:
        keep.append(torch.empty(padmb << 20, dtype=torch.uint8, device="cuda"))
    if k >= 8:
        keep.clear()
        torch.cuda.empty_cache()
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    for _ in range(6):
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    torch.cuda.synchronize()
    iters = 100
    b2b = [spmv_acc_amd.time_spmv_total(strat, iters, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / iters * 1e3 for _ in range(3)]
    reset = float(np.median(spmv_acc_amd.time_spmv(strat, 30, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))) * 1e3
    GiB = 1 << 30
    print(f"{name} {strat} pad {padmb:5d} MB: b2b " + " ".join(f"{t:.2f}" for t in b2b) + f"  reset {reset:.2f}   ci @ {ci.data_ptr() / GiB:9.4f} GiB  v @ {v.data_ptr() / GiB:9.4f}  x @ {x.data_ptr() / GiB:9.4f}  y @ {y.data_ptr() / GiB:9.4f}", flush=True)
    spmv_acc_amd.release_plans(rp)
    del rp, ci, v, x, y, y0
