#!/usr/bin/env python3
"""Round 5: the y reset of the per-launch protocol as a NON-TEMPORAL copy kernel (SPMV_ACC_RESET_NT, default since round 5) against the default-policy
copy of rounds 1-4 (SPMV_ACC_RESET_NT=0): event pair and kernel clock per stand-in, one process per setting.
    for nt in 0 1 0 1; do SPMV_ACC_RESET_NT=$nt python profiles/probes/reset_nt_probe.py Bump_2911; done
(profiles/r05_reset_copy_policy.txt)"""
import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch, spmv_acc_amd
from spmv_acc_amd import synth
name = sys.argv[1]
m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
x = torch.rand(n, device="cuda", dtype=torch.float64); y0 = torch.rand(m, device="cuda", dtype=torch.float64); y = y0.clone()
out = {}
for strat in ("flat", "adaptive"):
    ev, kn, ln = spmv_acc_amd.time_spmv_kernels(strat, 40, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)
    per = spmv_acc_amd.time_spmv(strat, 40, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)
    out[strat] = (round(float(np.median(per)) * 1e3, 2), round(float(np.median(kn)) * 1e3, 2))
print(name, os.environ.get("SPMV_ACC_RESET_NT", "default(nt)"), out)
