#!/usr/bin/env python3
"""What the per-launch protocol (y reset + one event pair per launch) costs before a byte of a matrix moves: a matrix WITHOUT non-zeros
(the library launches its y-scaling kernel only: no guard, no LDS, no barrier), a 256-row diagonal matrix through each kernel family,
with default events and with hipEventDisableSystemFence events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spmv_acc_amd

m = 256
x = torch.ones(m, dtype=torch.float64, device="cuda")
y0 = torch.zeros(m, dtype=torch.float64, device="cuda")
y = y0.clone()
empty = (torch.zeros(m + 1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.int32, device="cuda"), torch.zeros(1, dtype=torch.float64, device="cuda"), 0)
diag = (torch.arange(m + 1, dtype=torch.int32, device="cuda"), torch.arange(m, dtype=torch.int32, device="cuda"), torch.ones(m, dtype=torch.float64, device="cuda"), m)
for name, (rp, ci, v, nnz), strats in (("no non-zeros (y-scaling kernel only)", empty, ["line_enhance"]),
                                       ("256-row diagonal", diag, ["line_enhance", "flat", "adaptive_plus", "vector_row", "wf_row"])):
    for strat in strats:
        for _ in range(5):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, m, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
        out = []
        for flags in (0, spmv_acc_amd.EVENT_DISABLE_SYSTEM_FENCE):
            t = spmv_acc_amd.time_spmv(strat, 200, 1.0, 1.0, m, m, nnz, rp, ci, v, x, y, y0=y0, event_flags=flags)
            out.append((np.median(t) * 1e3, np.min(t) * 1e3))
        b2b = spmv_acc_amd.time_spmv_total(strat, 500, 1.0, 1.0, m, m, nnz, rp, ci, v, x, y) / 500 * 1e3
        print(f"{name:38s} {strat:14s} default events {out[0][0]:5.2f} us (min {out[0][1]:.2f})   fence-free events {out[1][0]:5.2f} us (min {out[1][1]:.2f})   back to back {b2b:5.2f} us", flush=True)
    spmv_acc_amd.release_plans(rp)
