#!/usr/bin/env python3
"""Row-length laws the named stand-ins do not cover (real SuiteSparse matrices are less regular than they are):
  lognormal SIGMA [MEAN]  row lengths ~ lognormal with the given sigma and mean (default 30); ROWLAW_M=rows (default 1.5 M)
  spikes K LEN [B]   rows of about B (default 30) non-zeros plus K rows of LEN non-zeros scattered through the matrix
  empty FRAC         rows of 20..40, a fraction FRAC of the rows empty
Per-launch hipEvent medians for the kernel families + what adaptive chose."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth

m = n = int(os.environ.get("ROWLAW_M", "1500000"))
g = torch.Generator(device="cuda"); g.manual_seed(9)
law = sys.argv[1]
if law == "lognormal":
    sigma = float(sys.argv[2])
    mean = float(sys.argv[3]) if len(sys.argv) >= 4 else 30.0
    lens = torch.exp(torch.randn(m, generator=g, device="cuda") * sigma + (np.log(mean) - sigma * sigma / 2)).long().clamp_(0, 2000000)
elif law == "spikes":
    k, length = int(sys.argv[2]), int(sys.argv[3])
    base = int(sys.argv[4]) if len(sys.argv) >= 5 else 30  # mean length of the ordinary rows
    lens = torch.randint(base - base // 6, base + base // 6 + 1, (m,), generator=g, device="cuda")
    lens[torch.randint(0, m, (k,), generator=g, device="cuda")] = length
else:
    frac = float(sys.argv[2])
    lens = torch.randint(20, 41, (m,), generator=g, device="cuda")
    lens[torch.rand(m, generator=g, device="cuda") < frac] = 0
rp = torch.zeros(m + 1, dtype=torch.int64, device="cuda"); torch.cumsum(lens, 0, out=rp[1:])
nnz = int(rp[-1].item())
rows = torch.repeat_interleave(torch.arange(m, device="cuda"), lens, output_size=nnz)
cols_law = os.environ.get("ROWLAW_COLS", "near")  # near: +-40 of the diagonal; clusters: three such windows n/3 apart; uniform: anywhere
if cols_law == "uniform":
    ci = torch.randint(0, n, (nnz,), generator=g, device="cuda")
else:
    ci = rows + torch.randint(-40, 41, (nnz,), generator=g, device="cuda")
    if cols_law == "clusters":
        ci = (ci + torch.randint(0, 3, (nnz,), generator=g, device="cuda") * (n // 3)) % n
    ci = ci.clamp_(0, n - 1)
key, _ = torch.sort(rows * n + ci); ci = (key % n).to(torch.int32); del key, rows
v = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
rp = rp.to(torch.int32)
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64); y = torch.zeros(m, dtype=torch.float64, device="cuda")
balg = synth.algorithmic_bytes(m, n, nnz)
print(f"{' '.join(sys.argv[1:])}: m={m} nnz={nnz} max row={int(lens.max())}  B_alg={balg/1e6:.0f} MB")
for strat in ("adaptive", "line_enhance", "adaptive_plus", "flat"):
    ms = spmv_acc_amd.time_spmv(strat, 25, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y)[5:]
    t = float(np.median(ms)) * 1e-3
    print(f"  {strat:14s} {t*1e6:8.1f} us  {balg/t/1e9:7.0f} GB/s")
