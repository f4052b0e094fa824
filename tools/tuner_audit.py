#!/usr/bin/env python3
"""Does the per-matrix tuner pick the fastest configuration?  For every sweep stand-in: the steady-state time of the plan the
library builds by itself against every pinned combination of the timed choices (cache policy x flat's cut-row form; cache policy
for the row blocks), all at alpha = beta = 1, back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

lib = spmv_acc_amd.load_library()
names = sys.argv[1:] or list(synth.SWEEP_NAMES)


def steady(strat, A, x, y, iters):
    m, n, nnz, rp, ci, v = A
    for _ in range(8):
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    return spmv_acc_amd.time_spmv_total(strat, iters, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / iters * 1e3


for name in names:
    A = synth.sweep_standin_torch(name)
    m, n, nnz, rp = A[0], A[1], A[2], A[3]
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.zeros(m, device="cuda", dtype=torch.float64)
    torch.cuda.synchronize()
    iters = 200 if nnz < 20_000_000 else 60
    for strat, grid in (("flat", [(p, f) for p in (0, 1, 3) for f in (0, 1)]), ("line_enhance", [(p, None) for p in (0, 1, 3)])):
        lib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans(rp)
        auto = steady(strat, A, x, y, iters)
        info = spmv_acc_amd.query_plan(rp, m)
        res = {}
        for pol, fin in grid:
            lib.spmv_acc_reset_tunables()
            lib.spmv_acc_set_tunable(b"stream_plain", pol)
            if fin is not None:
                lib.spmv_acc_set_tunable(b"flat_finish", fin)
            spmv_acc_amd.release_plans(rp)
            res[(pol, fin)] = steady(strat, A, x, y, iters)
        best = min(res, key=res.get)
        print(f"{name:18s} {strat:13s} auto {auto:7.2f} us (policy {info['stream_policy']}, fixup {info['flat_fixup']})   best pinned {res[best]:7.2f} us at {best}   "
              f"auto/best {auto / res[best]:.3f}   all: " + " ".join(f"{k}:{v:.1f}" for k, v in res.items()), flush=True)
    lib.spmv_acc_reset_tunables()
    spmv_acc_amd.release_plans(rp)
    del A, x, y
    torch.cuda.empty_cache()
