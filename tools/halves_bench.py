#!/usr/bin/env python3
"""A matrix whose two row halves differ in density (default 40 vs 5 nnz per row; argv: d0 d1) (adaptive.cpp:34-35 sends it to the two-width vector-row split): which kernel
family serves it best?  Per-launch hipEvent medians."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth

m = n = 2_000_000
d0, d1 = (int(a) for a in (sys.argv[1:3] if len(sys.argv) >= 3 else (40, 5)))  # mean nnz per row of the two halves
g = torch.Generator(device="cuda"); g.manual_seed(5)
stripe = int(sys.argv[3]) if len(sys.argv) >= 4 else 0  # 0: two halves; else alternate densities every `stripe` rows
lens = torch.cat([torch.randint(d0 - d0 // 4, d0 + d0 // 4 + 1, (m // 2,), generator=g, device="cuda"),
                  torch.randint(d1 - d1 // 4, d1 + d1 // 4 + 1, (m - m // 2,), generator=g, device="cuda")])
if stripe > 0:
    dense = ((torch.arange(m, device="cuda") // stripe) % 2) == 0
    perm = torch.cat([torch.nonzero(dense).flatten(), torch.nonzero(~dense).flatten()])  # first half of lens is the dense law
    lens = torch.empty_like(lens).scatter_(0, perm, lens)
rp = torch.zeros(m + 1, dtype=torch.int64, device="cuda"); torch.cumsum(lens, 0, out=rp[1:])
nnz = int(rp[-1].item())
rows = torch.repeat_interleave(torch.arange(m, device="cuda"), lens, output_size=nnz)
ci = ((rows + torch.randint(-40, 41, (nnz,), generator=g, device="cuda")).clamp_(0, n - 1)).to(torch.int32)
# sort columns within rows (CSR convention)
key = rows * n + ci.long(); key, _ = torch.sort(key); ci = (key % n).to(torch.int32)
v = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
rp = rp.to(torch.int32)
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64); y = torch.zeros(m, dtype=torch.float64, device="cuda")
balg = synth.algorithmic_bytes(m, n, nnz)
print(f"m={m} nnz={nnz} halves {int(rp[m//2])} / {nnz-int(rp[m//2])}  B_alg={balg/1e6:.0f} MB")
for strat in ("adaptive", "line_enhance", "adaptive_plus", "flat", "vector_row"):
    ms = spmv_acc_amd.time_spmv(strat, 25, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y)[5:]
    info = spmv_acc_amd.query_plan(rp, m)
    t = float(np.median(ms)) * 1e-3
    print(f"  {strat:14s} {t*1e6:8.1f} us  {balg/t/1e9:7.0f} GB/s   branch={info['adaptive_branch']}")
