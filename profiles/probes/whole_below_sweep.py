#!/usr/bin/env python3
"""slab_whole_below re-swept on R-MAT 24 / 25 / 26 now that the whole-row pass takes gather hints (the round-4 sweep without them:
8 / 16 / 24 / 32 / 48 / 64 / 128 / 256 -> 5.33 / 5.28 / 5.19 / 5.19 / 5.21 / 5.25 / 5.29 / 5.79 ms on R-MAT 25).
    python profiles/probes/whole_below_sweep.py [scales=24,25,26]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth

lib = spmv_acc_amd.load_library()
for scale in [int(t) for t in (sys.argv[1] if len(sys.argv) > 1 else "24,25,26").split(",")]:
    m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
    x = torch.rand(n, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    out = []
    for T in (16, 24, 32, 48, 64, 96, 128, 192):
        lib.spmv_acc_reset_tunables()
        lib.spmv_acc_set_tunable(b"slab_whole_below", T)
        spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="line_enhance")
        ms = min(float(np.median(spmv_acc_amd.time_spmv("line_enhance", 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))) for _ in range(2))
        out.append(f"{T}: {ms * 1e3:.0f} ({spmv_acc_amd.query_plan(rp, m)['slab_passes']})")
        spmv_acc_amd.release_plans(rp)
    print(f"R-MAT {scale}: slab_whole_below -> us (column slabs): " + "  ".join(out), flush=True)
    del rp, ci, v, x, y, y0
    torch.cuda.empty_cache()
lib.spmv_acc_reset_tunables()
