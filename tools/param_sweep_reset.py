#!/usr/bin/env python3
"""Tunable variants under the PER-LAUNCH protocol (y reset before each launch, one event pair per launch, median), interleaved
over 3 rounds, one fresh plan per variant and round:
    python tools/param_sweep_reset.py <strategy> <workload,...> "name=val,name=val;name=val;..." """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

lib = spmv_acc_amd.load_library()
strat, names, variants = sys.argv[1], sys.argv[2].split(","), sys.argv[3].split(";")
for name in names:
    if name == "banded":  # BASELINE configs[4]: rank 3's 32 M-row shard
        rows, total = 32_000_000, 256_000_000
        rp, ci, v = synth.banded_torch(rows, first_row=3 * rows, total_rows=total, device="cuda")
        m, n, nnz = rows, total, int(rp[-1].item())
    elif name.startswith("rmat"):
        m, n, nnz, rp, ci, v = synth.rmat_torch(int(name[4:]), device="cuda", seed=0xC4)
    else:
        m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    res = {var: [] for var in variants}
    b2b = {var: [] for var in variants}
    for rnd in range(3):
        for var in variants:
            lib.spmv_acc_reset_tunables()
            for kv in filter(None, var.split(",")):
                k, val = kv.split("=")
                assert lib.spmv_acc_set_tunable(k.encode(), int(val)) == 0, kv
            spmv_acc_amd.release_plans(rp)
            for _ in range(6):
                spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
            res[var].append(float(np.median(spmv_acc_amd.time_spmv(strat, 30, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))) * 1e3)
            iters = 200 if nnz < 20_000_000 else 60
            b2b[var].append(spmv_acc_amd.time_spmv_total(strat, iters, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / iters * 1e3)
    base = np.median(res[variants[0]])
    print(f"{name} {strat}: " + " | ".join(f"[{var or 'default'}] reset {np.median(res[var]):.2f} ({np.median(res[var]) / base:.3f}) b2b {np.median(b2b[var]):.2f}" for var in variants), flush=True)
    lib.spmv_acc_reset_tunables()
    spmv_acc_amd.release_plans(rp)
