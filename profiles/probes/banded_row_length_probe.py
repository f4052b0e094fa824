#!/usr/bin/env python3
"""Round 5: is the banded shard's kernel (one lane per row, all rows 8 long: every lane reads its row of the LDS tile at a 64-B stride, 8-way bank
conflicts -- profiles/r05_pmc_banded_shard.txt) held back by those conflicts?  The same 32 M-row band with 7, 8, 9, 10 and 12 offsets per row
(9 per row: a 72-B stride, conflict-free): algorithmic bytes / kernel time for each.  If 9 per row moves bytes markedly faster than 8, the conflicts matter."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import spmv_acc_amd
from spmv_acc_amd import synth

rows, total = 32_000_000, 256_000_000
for width, offs in ((7, tuple(range(-3, 4))), (8, tuple(range(-4, 4))), (9, tuple(range(-4, 5))), (10, tuple(range(-5, 5))), (12, tuple(range(-6, 6))),
                    (16, tuple(range(-8, 8))), (8, tuple(range(-4, 4)))):
    rp, ci, v = synth.banded_torch(rows, first_row=3 * rows, total_rows=total, device="cuda", offsets=offs)
    nnz = int(rp[-1].item())
    x = torch.rand(total, device="cuda", dtype=torch.float64)
    y0 = torch.rand(rows, device="cuda", dtype=torch.float64)
    y = y0.clone()
    out = []
    for strat in ("line_enhance", "flat"):
        ev, kn, ln = spmv_acc_amd.time_spmv_kernels(strat, 30, 1.0, 0.0, rows, total, nnz, rp, ci, v, x, y, y0=y0)
        b = synth.algorithmic_bytes(rows, rows + width, nnz, beta_nonzero=False)
        info = spmv_acc_amd.query_plan(rp, rows)
        out.append(f"{strat} {np.median(kn) * 1e3:7.1f} us kernel = {b / np.median(kn) / 1e9:5.2f} TB/s algorithmic ({info['last_kernel']}, vec {info['vec']}, policy {info['stream_policy']})")
    print(f"{width:2d} per row, {nnz / 1e6:.0f} M non-zeros: " + " | ".join(out), flush=True)
    spmv_acc_amd.release_plans(rp)
    del rp, ci, v, x, y, y0
    torch.cuda.empty_cache()
