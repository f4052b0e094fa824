#!/usr/bin/env python3
"""Back-to-back steady-state A/B of tunable variants (one fresh plan per variant):
    python tools/ab_steady.py <strategy> <workload,...> "name=val,name=val;name=val;..." """
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

lib = spmv_acc_amd.load_library()
strat, names, variants = sys.argv[1], sys.argv[2].split(","), sys.argv[3].split(";")
for name in names:
    if name.startswith("rmat"):  # rmat25 = BASELINE configs[3]
        m, n, nnz, rp, ci, v = synth.rmat_torch(int(name[4:]), device="cuda", seed=0xC4)
    else:
        m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.zeros(m, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    iters = 200 if nnz < 20_000_000 else (80 if nnz < 200_000_000 else 20)
    out = []
    first_y = {}
    for rnd in range(2):
        for var in variants:
            lib.spmv_acc_reset_tunables()
            for kv in filter(None, var.split(",")):
                k, val = kv.split("=")
                assert lib.spmv_acc_set_tunable(k.encode(), int(val)) == 0, kv
            spmv_acc_amd.release_plans(rp)
            for _ in range(8):
                spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
            t = spmv_acc_amd.time_spmv_total(strat, iters, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / iters * 1e3
            out.append((rnd, var, t))
            if rnd == 0:  # the variants are speed matters only: one SpMV from y = 0 must give the same vector under each
                y.zero_()
                spmv_acc_amd.csr_spmv(1.0, 0.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
                torch.cuda.synchronize()
                first_y[var] = y.clone()
                y.zero_()
    ref = first_y[variants[0]]
    scale = float(ref.abs().max()) or 1.0
    worst = max(float((first_y[var] - ref).abs().max()) / scale for var in variants)
    print(name, strat, f"max difference between variants {worst:.1e} |", " | ".join(f"[{var or 'default'}] " + " ".join(f"{t:.2f}" for r, v2, t in out if v2 == var) for var in variants), flush=True)
    lib.spmv_acc_reset_tunables()
    spmv_acc_amd.release_plans(rp)
