#!/usr/bin/env python3
"""How stable are adaptive's per-matrix timed choices?  N fresh plans on the same matrix in one process: the family and cache
policy each plan settled on and the steady-state time it then delivers."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
m, n, nnz = synth.LARGE_SET["Hardesty3"]
for tag, far in (("far 0.10", 0.10), ("far 0.00", 0.0)):
    rp, ci, v = synth.structured_csr_torch(m, n, nnz, 0xC2, device="cuda", far_fraction=far)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.zeros(m, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    for t in range(trials):
        spmv_acc_amd.release_plans(rp)
        ms = spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="adaptive")
        steady = spmv_acc_amd.time_spmv_total("adaptive", 100, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / 100
        info = spmv_acc_amd.query_plan(rp, m)
        print(f"{tag} trial {t}: family {info['adaptive_family']} policy {info['stream_policy']} prepare {ms:6.2f} ms steady {steady * 1e3:7.2f} us", flush=True)
    for strat in ("line_enhance", "adaptive_plus", "flat"):
        spmv_acc_amd.release_plans(rp)
        spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
        steady = spmv_acc_amd.time_spmv_total(strat, 100, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / 100
        print(f"{tag} {strat:14s} policy {spmv_acc_amd.query_plan(rp, m)['stream_policy']} steady {steady * 1e3:7.2f} us", flush=True)
    spmv_acc_amd.release_plans(rp)
