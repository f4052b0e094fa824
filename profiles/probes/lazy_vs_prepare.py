#!/usr/bin/env python3
"""Round 4's first-call budget finishes the per-matrix timings lazily over the following calls.  Does a plan settled that way end up as fast as one
settled by spmv_acc_prepare?  Per sweep stand-in and strategy: steady time (60 back-to-back launches between two events, best of 3) after
(A) prepare, (B) 40 plain calls; `settled` of both, and the choices query_plan reports.
    python profiles/probes/lazy_vs_prepare.py [name ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

names = sys.argv[1:] or list(synth.SWEEP_NAMES)


def steady(strat, A, x, y, iters=60):
    m, n, nnz, rp, ci, v = A
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


worst = 0.0
for name in names:
    A = synth.sweep_standin_torch(name)
    m, n, nnz, rp, ci, v = A
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.zeros(m, device="cuda", dtype=torch.float64)
    for strat in ("adaptive", "flat", "adaptive_plus"):
        row = []
        for mode in ("prepare", "40 calls", "prepare", "40 calls"):
            spmv_acc_amd.release_plans(rp)
            if mode == "prepare":
                spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
            else:
                for _ in range(40):
                    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
                torch.cuda.synchronize()
            info = spmv_acc_amd.query_plan(rp, m)
            t = steady(strat, A, x, y)
            row.append((mode, t, info["settled"], info["stream_policy"], info["adaptive_family"]))
        a = min(r[1] for r in row if r[0] == "prepare")
        b = min(r[1] for r in row if r[0] != "prepare")
        worst = max(worst, b / a - 1)
        print(f"{name:18s} {strat:14s} " + "  ".join(f"{r[0]}: {r[1]:7.2f} us settled={r[2]} pol={r[3]} fam={r[4]}" for r in row) + f"   lazy / prepare = {b / a:.3f}", flush=True)
    spmv_acc_amd.release_plans(rp)
    del A, rp, ci, v, x, y
    torch.cuda.empty_cache()
print(f"worst lazy / prepare - 1 = {worst * 100:.1f} %")
