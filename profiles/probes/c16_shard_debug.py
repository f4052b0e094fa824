#!/usr/bin/env python3
"""Round 6 debugging aid: the un-rebased row shard of tests/test_gpu_parity.py::test_row_shard_without_rebasing, one strategy at a time with the
16-bit column encoding forced, printing before every call (the call that aborted is the last line)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth
lib = spmv_acc_amd.load_library()
lib.spmv_acc_set_tunable(b"col16", int(os.environ.get("C16", "1")))
rowptr, cols, vals = synth.random_csr(30000, 30000, 11, seed=4, kind="powerlaw")
rng = np.random.default_rng(1)
x, y0 = rng.standard_normal(30000), rng.standard_normal(30000)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
drp, dci, dv, dx = d(rowptr), d(cols), d(vals), d(x)
strategies = sys.argv[1].split(",") if len(sys.argv) > 1 else ["line_enhance", "flat"]
for r0, r1 in ((12345, 27001), (1, 30000), (29990, 30000), (7, 8)):
    for strat in strategies:
        print("call", r0, r1, strat, "nnz0", int(rowptr[r0]), "nnz", int(rowptr[r1]), flush=True)
        dy = d(y0)
        spmv_acc_amd.csr_spmv(1.0, 1.0, r1 - r0, 30000, int(rowptr[r1]), drp[r0:], dci, dv, dx, dy[r0:], strategy=strat)
        torch.cuda.synchronize()
        print("  ok", spmv_acc_amd.query_plan(drp[r0:], r1 - r0), flush=True)
        spmv_acc_amd.release_plans(drp[r0:])
print("done")
