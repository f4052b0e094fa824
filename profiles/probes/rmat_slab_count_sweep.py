#!/usr/bin/env python3
"""R-MAT 24 / 25 / 26 under forced slab counts around what the automatic rule picks (slabs of ~32 MB of x): is the rule's S still the best S
with the two-class lists and the wavefront merge?    python profiles/probes/rmat_slab_count_sweep.py [scales=24,25,26]"""
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
    auto = max(2, min(16, (n * 8 + (16 << 20)) // (32 << 20)))
    out = []
    for S in sorted({max(2, auto // 2), max(2, auto - 3), max(2, auto - 2), max(2, auto - 1), auto, min(16, auto + 1), min(16, auto + 2), min(16, auto + 4), 0}):
        lib.spmv_acc_reset_tunables()
        lib.spmv_acc_set_tunable(b"slab_segments", S)
        spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="line_enhance")
        ms = min(float(np.median(spmv_acc_amd.time_spmv("line_enhance", 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))) for _ in range(2))
        out.append(f"S={S}: {ms * 1e3:.0f}")
        spmv_acc_amd.release_plans(rp)
    print(f"R-MAT {scale} (x {n * 8 >> 20} MB, rule S = {auto}): " + "  ".join(out) + "  us", flush=True)
    del rp, ci, v, x, y, y0
    torch.cuda.empty_cache()
lib.spmv_acc_reset_tunables()
