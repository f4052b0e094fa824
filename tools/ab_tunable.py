#!/usr/bin/env python3
"""In-process sweep of ONE tunable on the same device arrays: for each value the plan is rebuilt (prepare), then the per-launch protocol's kernel clock
and the back-to-back mean are taken; `rounds` passes over the values, medians reported.
usage: tools/ab_tunable.py <strategy> <workload,...> <tunable> <v0,v1,...> [rounds] [fixed tunables k=v,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

strat, names, knob = sys.argv[1], sys.argv[2].split(","), sys.argv[3]
values = [int(v) for v in sys.argv[4].split(",")]
rounds = int(sys.argv[5]) if len(sys.argv) > 5 else 3
fixed = [kv.split("=") for kv in (sys.argv[6].split(",") if len(sys.argv) > 6 and sys.argv[6] else [])]
lib = spmv_acc_amd.load_library()
for name in names:
    if name == "banded":
        m, n = 32_000_000, 256_000_000
        rp, ci, v = synth.banded_torch(m, first_row=3 * m, total_rows=n, device="cuda")
        nnz = int(rp[-1].item())
    else:
        m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    balg = synth.algorithmic_bytes(m, n, nnz)
    res = {val: {"kernel": [], "b2b": []} for val in values}
    for rnd in range(rounds):
        for val in (values if rnd % 2 == 0 else values[::-1]):
            lib.spmv_acc_reset_tunables()
            for k, f in fixed:
                assert lib.spmv_acc_set_tunable(k.encode(), int(f)) == 0
            assert lib.spmv_acc_set_tunable(knob.encode(), val) == 0
            spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
            _, kn, _ = spmv_acc_amd.time_spmv_kernels(strat, 30, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)
            iters = 200 if nnz < 20_000_000 else 60
            b2b = spmv_acc_amd.time_spmv_total(strat, iters, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / iters
            res[val]["kernel"].append(float(np.median(kn)) * 1e3)
            res[val]["b2b"].append(b2b * 1e3)
            y.copy_(y0)
            spmv_acc_amd.release_plans(rp)
    line = f"{name:16s} {strat:12s} {knob}:"
    for val in values:
        k, b = float(np.median(res[val]["kernel"])), float(np.median(res[val]["b2b"]))
        line += f" | {val}: kernel {k:.2f} ({balg / k / 8e6:.3f}) b2b {b:.2f} ({balg / b / 8e6:.3f})"
    print(line, flush=True)
lib.spmv_acc_reset_tunables()
