#!/usr/bin/env python3
"""Round 4: the full-size checks of tests/test_gpu_configs.py (device reference, alpha / beta decomposition, row sums, linearity, a row prefix
against the CPU oracle) on R-MAT scale 26 -- 67 M rows, ~1.05 G non-zeros: past 2^30 non-zeros and 2^32 bytes of every array with a power-law
matrix (the suite's R-MAT 25 stays 1.5 % below both; the int32-limit test is banded) -- under the automatic choice, the forced slab passes and the
one-kernel path, with per-SpMV times.
    python profiles/probes/rmat26_check.py [scale=26]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth
import oracle_lib
import test_gpu_configs as T

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 26
oracle_lib.lib()
A = synth.rmat_torch(scale, device="cuda", seed=0xC4)
m, n, nnz, rp, ci, v = A
lens = rp[1:] - rp[:-1]
print(f"R-MAT {scale}: m {m} nnz {nnz} ({nnz / 2**30:.3f} x 2^30), longest row {int(lens.max().item())}, values array {nnz * 8 / 2**32:.2f} x 2^32 bytes", flush=True)
del lens
lib = spmv_acc_amd.load_library()
x = torch.rand(n, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, device="cuda", dtype=torch.float64)
for name, knobs, strats in (("automatic", {}, ("line_enhance", "flat", "adaptive", "adaptive_plus")),
                            ("slab passes forced (8)", {"slab_segments": 8}, ("line_enhance",)),
                            ("one-kernel path", {"slab_segments": 0}, ("line_enhance", "vector_row"))):
    lib.spmv_acc_reset_tunables()
    for k, val in knobs.items():
        assert lib.spmv_acc_set_tunable(k.encode(), val) == 0
    print(f"-- {name}", flush=True)
    T._full_size_checks(torch, oracle_lib, A, strats, seed=0xC4C6)
    for strat in strats:
        y = y0.clone()
        ms = float(np.median(spmv_acc_amd.time_spmv(strat, 6, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)))
        info = spmv_acc_amd.query_plan(rp, m)
        print(f"   {strat:14s} {ms * 1e3:9.1f} us   slab_passes {info['slab_passes']}  plus_blocks {info['plus_blocks']}", flush=True)
    spmv_acc_amd.release_plans(rp)
lib.spmv_acc_reset_tunables()
print("all checks passed", flush=True)
