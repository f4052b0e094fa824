#!/usr/bin/env python3
"""First-call (plan) cost per strategy on the sweep stand-ins: spmv_acc_prepare's device time and the steady-state launch,
i.e. how many SpMVs the one-time structural passes + per-matrix timings are worth -- and the same first call when the timed
choices come from the tune cache (SPMV_ACC_TUNE_CACHE / spmv_acc_set_tune_cache: what a second process pays)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

cache = os.path.join(tempfile.mkdtemp(), "tune.txt")
spmv_acc_amd.set_tune_cache(cache)
for name in (sys.argv[1:] or ["Hardesty3", "Bump_2911", "largebasis", "scircuit"]):
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.zeros(m, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    for strat in ("line_enhance", "flat", "adaptive", "adaptive_plus", "vector_row"):
        spmv_acc_amd.release_plans(rp)
        ms = spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
        steady = spmv_acc_amd.time_spmv_total(strat, 50, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / 50
        spmv_acc_amd.release_plans(rp)  # a fresh plan, as a second process would build it: choices adopted from the cache
        ms2 = spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
        print(f"{name:12s} {strat:14s} prepare {ms:8.3f} ms   steady {steady * 1e3:8.2f} us   = {ms / steady:6.0f} SpMVs"
              f"   with cached choices {ms2:7.3f} ms = {ms2 / steady:5.0f} SpMVs", flush=True)
    spmv_acc_amd.release_plans(rp)
