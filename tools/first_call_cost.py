#!/usr/bin/env python3
"""What the FIRST call on a matrix costs since round 4 (tunables first_call_budget / later_call_budget): wall time of call 1 (the bounded one), of calls
2..12 (each may resume the open timings with a small budget), how many calls it takes until every timing is settled, the steady-state launch after
that, the same with an unbounded first call (first_call_budget = 0: rounds 1-3), and spmv_acc_prepare (everything up front).
    python tools/first_call_cost.py [stand-in ... | rmat25]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

lib = spmv_acc_amd.load_library()


def wall_calls(strat, A, x, y, ncalls):
    m, n, nnz, rp, ci, v = A
    out, prep = [], []
    for _ in range(ncalls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
        prep.append(lib.spmv_acc_last_prepare_us() * 1e-3)
    return out, prep


for name in (sys.argv[1:] or ["Hardesty3", "Bump_2911", "largebasis", "scircuit", "rmat25"]):
    if name.startswith("rmat"):
        A = synth.rmat_torch(int(name[4:]), device="cuda", seed=0xC4)
        strategies = ("line_enhance",)
    else:
        A = synth.sweep_standin_torch(name)
        strategies = ("adaptive", "flat", "line_enhance")
    m, n, nnz, rp, ci, v = A
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.zeros(m, device="cuda", dtype=torch.float64)
    # (one throw-away call on another matrix first: code-object load and allocator warm-up are not this matrix's cost)
    tm = 1024
    trp = torch.arange(tm + 1, dtype=torch.int32, device="cuda"); tci = torch.arange(tm, dtype=torch.int32, device="cuda")
    for s_ in strategies:
        spmv_acc_amd.csr_spmv(1.0, 1.0, tm, tm, tm, trp, tci, torch.ones(tm, dtype=torch.float64, device="cuda"), torch.ones(tm, dtype=torch.float64, device="cuda"),
                              torch.zeros(tm, dtype=torch.float64, device="cuda"), strategy=s_)
    torch.cuda.synchronize()
    for strat in strategies:
        rows = {}
        for label, first_budget in (("bounded (default)", None), ("unbounded first call (rounds 1-3)", 0)):
            lib.spmv_acc_reset_tunables()
            if first_budget is not None:
                lib.spmv_acc_set_tunable(b"first_call_budget", first_budget)
                lib.spmv_acc_set_tunable(b"later_call_budget", 0)
            spmv_acc_amd.release_plans(rp)
            calls, prep = wall_calls(strat, A, x, y, 40)
            steady = float(np.median(spmv_acc_amd.time_spmv(strat, 20, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y.clone()))) 
            settled = max([i for i, p_ in enumerate(prep) if p_ > 0.0] + [0]) + 1
            rows[label] = (calls, prep, steady, settled)
            print(f"{name:12s} {strat:13s} {label:34s}: call 1 {calls[0]:8.3f} ms = {calls[0] / steady:6.1f} SpMVs; calls 2-6 {' '.join(f'{c:7.3f}' for c in calls[1:6])} ms; "
                  f"settled after call {settled:2d}; all plan work {sum(prep):8.3f} ms; steady {steady * 1e3:8.2f} us", flush=True)
        lib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans(rp)
        ms = spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
        calls, prep = wall_calls(strat, A, x, y, 3)
        print(f"{name:12s} {strat:13s} {'spmv_acc_prepare (everything up front)':34s}: {ms:8.3f} ms; plan work left for the calls after it: {sum(prep):.3f} ms", flush=True)
    spmv_acc_amd.release_plans(rp)
    del A, rp, ci, v, x, y
    torch.cuda.empty_cache()
lib.spmv_acc_reset_tunables()
