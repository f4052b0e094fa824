#!/usr/bin/env python3
"""BASELINE.json configs[2]: the large-set sweep (examples/large-data-set-batch.sh:24-52 dims, synthetic stand-ins)
+ scircuit-like and af_shell10-like, flat and adaptive strategies, with the reference's verdict (verify_y) on a
row sample and a rocSPARSE dcsrmv row (comparison only, as benchmark/benchmark_rocsparse.hpp does).  Prints a markdown table."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import spmv_acc_amd
from rocsparse_row import RocsparseCsrmv
from spmv_acc_amd import synth

EXTRA = synth.LARGE_SET_EXTRA  # SuiteSparse dims (stand-ins)
names = list(synth.LARGE_SET) + list(EXTRA)
ROC = RocsparseCsrmv()
print("| matrix (stand-in) | rows | nnz | nnz/row | strategy | us (median) | GFLOP/s | B_alg GB/s | frac of 8 TB/s | ref GiB/s | rows failing verify_y rule | rocSPARSE dcsrmv us: no analysis / with analysis |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|")
for i, name in enumerate(names):
    if name in EXTRA:
        m, n, nnz = EXTRA[name]
        rp, ci, v = synth.structured_csr_torch(m, n, nnz, 0xC30A + i, device="cuda", far_fraction=0.02, spread=[1, 1, 1, 2])
    elif name == "Hardesty3":
        m, n, nnz, rp, ci, v = synth.hardesty3_like_torch(device="cuda")
    else:
        m, n, nnz, rp, ci, v = synth.large_set_like_torch(name, device="cuda", seed=0xC300 + i)
    g = torch.Generator(device="cuda"); g.manual_seed(7)
    x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    rows = torch.repeat_interleave(torch.arange(m, device="cuda"), (rp[1:] - rp[:-1]).long(), output_size=nnz)
    ref = y0.clone().index_add_(0, rows, v * x[ci.long()])  # independent evaluation: y0 + sum_j v_j * x[col_j], fp64 on the GPU
    del rows
    try:
        yr = y0.clone()
        t_plain, _ = ROC.time(m, n, nnz, rp, ci, v, x, yr, analysis=False)
        t_adapt, t_an = ROC.time(m, n, nnz, rp, ci, v, x, yr, analysis=True)
        rs = f"{t_plain:.1f} / {t_adapt:.1f} (+{t_an:.0f} analysis)"
    except Exception as ex:  # noqa: BLE001
        rs = f"n/a ({type(ex).__name__})"
    balg = synth.algorithmic_bytes(m, n, nnz)
    for strat in ("flat", "adaptive"):
        y = y0.clone()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        d = (y - ref).abs()
        bad = torch.where(ref.abs() <= 1e-12, d >= 1e-14, d / ref.abs() >= 1e-7)  # verify_y's rule (cli/verification.cpp:15-38)
        failed = int(bad.sum().item())
        ms = spmv_acc_amd.time_spmv(strat, 30, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)[3:]
        med = float(np.median(ms)) * 1e-3
        print(f"| {name} | {m} | {nnz} | {nnz / m:.2f} | {strat} | {med * 1e6:.1f} | {2 * nnz / med / 1e9:.1f} | {balg / med / 1e9:.0f} | "
              f"{balg / med / 8e12:.3f} | {synth.reference_bytes(m, nnz) / 2**30 / med:.0f} | {failed} | {rs} |")
        sys.stdout.flush()
    spmv_acc_amd.release_plans()
    del rp, ci, v, x, y, y0
    torch.cuda.empty_cache()
