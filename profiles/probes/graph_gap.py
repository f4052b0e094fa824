#!/usr/bin/env python3
"""Back-to-back launches vs one hipGraph holding the same launches, on the small / medium sweep stand-ins: what the
inter-kernel gap costs a launch-bound SpMV loop (steady-state calls are launches only, so they capture)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

N = 200
side = torch.cuda.Stream()
for name in (sys.argv[1:] or ["scircuit", "largebasis", "Ga41As41H72", "TSOPF_RS_b2383"]):
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    g = torch.Generator(device="cuda"); g.manual_seed(1)
    x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    y = torch.rand(m, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    balg = synth.algorithmic_bytes(m, n, nnz)
    torch.cuda.synchronize()  # the generators ran on the default stream, the launches below go to `side`
    for strat in ("flat", "adaptive"):
        with torch.cuda.stream(side):
            for _ in range(10):
                spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
            side.synchronize()
            loop_us = spmv_acc_amd.time_spmv_total(strat, N, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / N * 1e3
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(N):
                spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        graph.replay(); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            e0.record(); graph.replay(); e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / N * 1e3)
        print(f"{name:16s} {strat:9s} loop {loop_us:7.2f} us ({balg / loop_us / 8e6:.3f})   graph {best:7.2f} us ({balg / best / 8e6:.3f})", flush=True)
    spmv_acc_amd.release_plans(rp)
    spmv_acc_amd.load_library().spmv_acc_set_stream(None)
