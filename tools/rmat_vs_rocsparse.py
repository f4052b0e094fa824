#!/usr/bin/env python3
"""BASELINE configs[3] / [4] beside rocSPARSE dcsrmv (comparison only, as benchmark/benchmark_rocsparse.hpp does for the sweep):
R-MAT scale 24 / 25 under line_enhance (default path, and with the column-slab passes off) and the 32 M-row banded shard under adaptive,
per-launch protocol with y reset.  Prints one line per matrix.
    python tools/rmat_vs_rocsparse.py [24 25 banded]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import spmv_acc_amd
from rocsparse_row import RocsparseCsrmv
from spmv_acc_amd import synth

ROC = RocsparseCsrmv()
lib = spmv_acc_amd.load_library()
for what in (sys.argv[1:] or ["24", "25", "banded"]):
    if what == "banded":
        rows, total = 32_000_000, 256_000_000
        rp, ci, v = synth.banded_torch(rows, first_row=3 * rows, total_rows=total, device="cuda")
        m, n, nnz, strat, beta = rows, total, int(rp[-1].item()), "adaptive", 0.0
    else:
        m, n, nnz, rp, ci, v = synth.rmat_torch(int(what), device="cuda", seed=0xC4)
        strat, beta = "line_enhance", 1.0
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    try:
        yr = y0.clone()
        t_plain, _ = ROC.time(m, n, nnz, rp, ci, v, x, yr, analysis=False)
        t_adapt, t_an = ROC.time(m, n, nnz, rp, ci, v, x, yr, analysis=True)
        rs = f"rocSPARSE dcsrmv {t_plain:.1f} us without / {t_adapt:.1f} us with analysis (+{t_an / 1e3:.1f} ms analysis)"
    except Exception as ex:  # noqa: BLE001
        rs = f"rocSPARSE n/a ({type(ex).__name__}: {ex})"
    out = []
    for tag, knobs in (("default", {}), ("slab_segments=0", {"slab_segments": 0})):
        if what == "banded" and knobs:
            continue
        lib.spmv_acc_reset_tunables()
        for k, val in knobs.items():
            lib.spmv_acc_set_tunable(k.encode(), val)
        y = y0.clone()
        for _ in range(3):
            spmv_acc_amd.csr_spmv(1.0, beta, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        ms = float(np.median(spmv_acc_amd.time_spmv(strat, 10, 1.0, beta, m, n, nnz, rp, ci, v, x, y, y0=y0)))
        out.append(f"{strat} [{tag}] {ms * 1e3:.1f} us")
        spmv_acc_amd.release_plans(rp)
    lib.spmv_acc_reset_tunables()
    print(f"{what}: m {m} nnz {nnz}: " + ", ".join(out) + f"; {rs}", flush=True)
    del rp, ci, v, x, y0
    torch.cuda.empty_cache()
