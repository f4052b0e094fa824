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
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.zeros(m, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    iters = 200 if nnz < 20_000_000 else 80
    out = []
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
    print(name, strat, " | ".join(f"[{var or 'default'}] " + " ".join(f"{t:.2f}" for r, v2, t in out if v2 == var) for var in variants), flush=True)
    lib.spmv_acc_reset_tunables()
    spmv_acc_amd.release_plans(rp)
