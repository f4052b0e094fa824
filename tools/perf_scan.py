#!/usr/bin/env python3
"""Hunt for slow corners: random row-length laws with columns near the diagonal (so gathers are cheap and anything far below the
streaming rate is the kernels' doing), adaptive + the three fixed families timed; prints the cases sorted by adaptive's B_alg GB/s.
    python tools/perf_scan.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import spmv_acc_amd
from spmv_acc_amd import synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
g = torch.Generator(device="cuda"); g.manual_seed(int(rng.integers(1 << 30)))
out = []
for case in range(cases):
    m = int(10 ** rng.uniform(float(os.environ.get("PERF_SCAN_LOG_M_MIN", "4.5")), 7.3))
    mean = float(rng.choice([1.5, 4, 12, 40, 150, 1000]))
    m = max(1000, min(m, int(1.2e8 / mean)))
    law = rng.choice(["lognormal", "spikes", "stripes", "empty", "bimodal", "ramp"])
    if law == "lognormal":
        sigma = float(rng.uniform(0, 2.2))
        lens = torch.exp(torch.randn(m, generator=g, device="cuda") * sigma + (np.log(mean) - sigma * sigma / 2)).long()
        desc = f"sigma={sigma:.2f}"
    elif law == "spikes":
        lens = torch.randint(0, int(2 * mean) + 1, (m,), generator=g, device="cuda")
        k, L = int(rng.integers(1, max(2, m // 100))), int(rng.choice([100, 600, 3000, 20000, 300000]))
        lens[torch.randint(0, m, (k,), generator=g, device="cuda")] = L
        desc = f"{k}x{L}"
    elif law == "stripes":
        stripe = int(rng.choice([3, 40, 700, 20000]))
        dense = ((torch.arange(m, device="cuda") // stripe) % 2) == 0
        ratio = float(rng.choice([2, 5, 20]))
        lens = torch.where(dense, torch.randint(0, int(2 * mean * ratio) + 1, (m,), generator=g, device="cuda"),
                           torch.randint(0, int(2 * mean) + 1, (m,), generator=g, device="cuda"))
        desc = f"stripe={stripe} x{ratio}"
    elif law == "empty":
        frac = float(rng.uniform(0.2, 0.995))
        lens = torch.randint(0, int(2 * mean) + 1, (m,), generator=g, device="cuda")
        lens[torch.rand(m, generator=g, device="cuda") < frac] = 0
        desc = f"empty={frac:.3f}"
    elif law == "bimodal":
        hi = int(rng.choice([64, 256, 2048, 5000]))
        p = float(rng.choice([0.001, 0.01, 0.1, 0.5]))
        lens = torch.where(torch.rand(m, generator=g, device="cuda") < p, torch.full((m,), hi, device="cuda"),
                           torch.randint(0, int(2 * mean) + 1, (m,), generator=g, device="cuda"))
        desc = f"{p}@{hi}"
    else:  # ramp: row length grows linearly through the matrix
        lens = (torch.arange(m, device="cuda").double() * (2 * mean / m)).long()
        desc = "0..2mean"
    total = int(lens.sum().item())
    if total > 1.3e8:
        lens = (lens.double() * (1.2e8 / total)).long()
    rp = torch.zeros(m + 1, dtype=torch.int64, device="cuda"); torch.cumsum(lens, 0, out=rp[1:])
    nnz = int(rp[-1].item())
    if nnz < 200000:
        continue
    n = m
    rows = torch.repeat_interleave(torch.arange(m, device="cuda"), lens, output_size=nnz)
    ci = (rows + torch.randint(-40, 41, (nnz,), generator=g, device="cuda")).clamp_(0, n - 1).to(torch.int32)
    del rows
    v = torch.rand(nnz, generator=g, device="cuda", dtype=torch.float64)
    x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64); y = torch.zeros(m, dtype=torch.float64, device="cuda")
    rp32 = rp.to(torch.int32)
    balg = synth.algorithmic_bytes(m, n, nnz)
    res = {}
    for strat in ("adaptive", "line_enhance", "adaptive_plus", "flat"):
        ms = spmv_acc_amd.time_spmv(strat, 12, 1.0, 1.0, m, n, nnz, rp32, ci, v, x, y)[4:]
        res[strat] = balg / (float(np.median(ms)) * 1e-3) / 1e9
    out.append((res["adaptive"], f"m={m} nnz={nnz} avg={nnz/m:.1f} max={int(lens.max())} {law} {desc}", res))
    print(f"case {case}: {out[-1][1]} -> adaptive {res['adaptive']:.0f} GB/s", flush=True)
    spmv_acc_amd.release_plans(rp32)
    del rp, rp32, ci, v, x, y, lens
    torch.cuda.empty_cache()
print("\n== sorted by adaptive GB/s (B_alg) ==")
for a, d, res in sorted(out, key=lambda t: t[0]):
    print(f"{a:7.0f}  [row blocks {res['line_enhance']:.0f} | plus {res['adaptive_plus']:.0f} | flat {res['flat']:.0f}]  {d}")
