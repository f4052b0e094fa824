#!/usr/bin/env python3
"""Driver for the counter passes over the FEM-class kernels of the configs[2] sweep (tools/pmc_fem.sh): one stand-in, one strategy, the plan
settled first, then `iters` launches under the reference harness's per-launch protocol (y restored by a device copy before every launch).
Prints one JSON line: the plan the library settled on (so that the shell script can pin it for the perturbed counter passes), the algorithmic
bytes and the per-launch / back-to-back event times of THIS process."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = argparse.ArgumentParser()
p.add_argument("--workload", default="Bump_2911")
p.add_argument("--strategy", default="flat")
p.add_argument("--iters", type=int, default=12)
p.add_argument("--no-timing", action="store_true", help="counter passes: launches only (no event pairs of our own)")
a = p.parse_args()
import numpy as np
import torch

import spmv_acc_amd
from spmv_acc_amd import synth

beta = 1.0
if a.workload == "banded_shard":  # BASELINE configs[4]: rank 3's 32 M-row shard of the 256 M-row banded matrix, beta = 0 (bench.py leg_banded_shard)
    m, n = 32_000_000, 256_000_000
    rp, ci, v = synth.banded_torch(m, first_row=3 * m, total_rows=n, device="cuda")
    nnz = int(rp[-1].item())
    beta = 0.0
    balg = synth.algorithmic_bytes(m, m + 7, nnz, beta_nonzero=False)
elif a.workload == "rmat25":  # BASELINE configs[3]
    m, n, nnz, rp, ci, v = synth.rmat_torch(25, device="cuda", seed=0xC4)
    balg = synth.algorithmic_bytes(m, n, nnz)
else:
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(a.workload, device="cuda")
    balg = synth.algorithmic_bytes(m, n, nnz)
g = torch.Generator(device="cuda")
g.manual_seed(1234)
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
y = y0.clone()
spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=a.strategy, beta=beta)
out = {"workload": a.workload, "strategy": a.strategy, "m": m, "n": n, "nnz": nnz, "algorithmic_bytes": balg}
if a.no_timing:
    for _ in range(a.iters):
        y.copy_(y0)
        spmv_acc_amd.csr_spmv(1.0, beta, m, n, nnz, rp, ci, v, x, y, strategy=a.strategy)
    torch.cuda.synchronize()
else:
    ev, kn, ln = spmv_acc_amd.time_spmv_kernels(a.strategy, a.iters, 1.0, beta, m, n, nnz, rp, ci, v, x, y, y0=y0)
    b2b = spmv_acc_amd.time_spmv_total(a.strategy, a.iters, 1.0, beta, m, n, nnz, rp, ci, v, x, y) / a.iters
    out.update(per_launch_us_median=round(float(np.median(ev)) * 1e3, 2), kernel_clock_us_median=round(float(np.median(kn)) * 1e3, 2),
               back_to_back_us=round(b2b * 1e3, 2))
out["plan"] = spmv_acc_amd.query_plan(rp, m)
print(json.dumps(out))
