#!/usr/bin/env python3
"""Times the row-block preprocessing passes (plan-time work): device vs host form of the adaptive-plus
analysis, the break-point kernel, and the first-call (plan-building) cost of each strategy."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth

for w in ("hardesty3", "rmat22"):
    if w == "hardesty3":
        m, n, nnz, rp, ci, v = synth.hardesty3_like_torch(device="cuda")
    else:
        m, n, nnz, rp, ci, v = synth.rmat_torch(22, device="cuda")
    hrp = rp.cpu().numpy()
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.zeros(m, device="cuda", dtype=torch.float64)
    spmv_acc_amd.adaptive_plus_analyze_device(rp, m, nnz, 1024, 256, 1)  # warm-up (rocPRIM init)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); b, _, _ = spmv_acc_amd.adaptive_plus_analyze_device(rp, m, nnz, 1024, 256, 1); torch.cuda.synchronize()
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter(); hb, _, _ = spmv_acc_amd.adaptive_plus_analyze(hrp, m, 1024, 256, 1); t_host = time.perf_counter() - t0
    t0 = time.perf_counter(); _ = rp.cpu(); t_d2h = time.perf_counter() - t0
    print(f"{w}: m={m} nnz={nnz} blocks={b} (host form {hb}) | analysis device {t_dev*1e3:.2f} ms | host loop {t_host*1e3:.2f} ms "
          f"(+ rowptr D2H {t_d2h*1e3:.2f} ms when no host copy exists)")
    n_bp = spmv_acc_amd.break_points_len(nnz, 2048)
    out = torch.empty(n_bp, dtype=torch.int32, device="cuda")
    spmv_acc_amd.break_points(rp, m, nnz, 2048, out); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); spmv_acc_amd.break_points(rp, m, nnz, 2048, out); e1.record(); e1.synchronize()
    print(f"   break points ({n_bp} entries, stride 2048): {e0.elapsed_time(e1)*1e3:.1f} us")
    for s in ("adaptive", "flat", "line_enhance", "adaptive_plus"):
        spmv_acc_amd.release_plans()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=s); torch.cuda.synchronize()
        first = time.perf_counter() - t0
        t0 = time.perf_counter(); spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=s); torch.cuda.synchronize()
        second = time.perf_counter() - t0
        print(f"   {s:14s} first call (plan + SpMV) {first*1e3:8.3f} ms   steady call {second*1e3:8.3f} ms")
