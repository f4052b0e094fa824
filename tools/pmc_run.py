#!/usr/bin/env python3
"""Small driver for rocprofv3 --pmc passes: a few launches of one strategy on one workload."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
p = argparse.ArgumentParser()
p.add_argument("--workload", default="hardesty3")
p.add_argument("--strategy", default="adaptive")
p.add_argument("--iters", type=int, default=5)
p.add_argument("--far", type=float, default=None)
a = p.parse_args()
import torch
import spmv_acc_amd
from spmv_acc_amd import synth
if a.workload == "hardesty3" and a.far is not None:
    m, n, nnz = synth.LARGE_SET["Hardesty3"]
    rp, ci, v = synth.structured_csr_torch(m, n, nnz, 0xC2, device="cuda", far_fraction=a.far)
elif a.workload == "hardesty3":
    m, n, nnz, rp, ci, v = synth.hardesty3_like_torch(device="cuda")
else:
    m, n, nnz, rp, ci, v = synth.large_set_like_torch(a.workload, device="cuda")
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
y = torch.rand(m, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
for _ in range(a.iters):
    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=a.strategy)
torch.cuda.synchronize()
print("done", m, n, nnz, synth.algorithmic_bytes(m, n, nnz))
