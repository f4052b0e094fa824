#!/usr/bin/env python3
"""The row-count and non-zero-count limits at once: 2,147,400,000 rows with one non-zero each (+ one row of 5001), nnz = 2,147,405,000 (the int32
tile arithmetic allows INT_MAX - 65536), y in closed form -- every strategy that makes sense on one-element rows, and the run-list build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, spmv_acc_amd
lib = spmv_acc_amd.load_library()
m, n, extra = 2_147_400_000, 1 << 20, 5000
rp = torch.arange(m + 1, dtype=torch.int32, device="cuda")
rp[m // 2 + 1:] += extra
nnz = m + extra
ci = torch.empty(nnz, dtype=torch.int32, device="cuda")
for a in range(0, nnz, 1 << 28):
    b = min(nnz, a + (1 << 28))
    ci[a:b] = ((torch.arange(a, b, dtype=torch.int64, device="cuda") * 7919) % n).to(torch.int32)
v = torch.ones(nnz, dtype=torch.float64, device="cuda")
x = torch.arange(n, dtype=torch.float64, device="cuda") % 13.0 - 6.0
def expected(rows):
    j = torch.where(rows > m // 2, rows + extra, rows)
    return 2.0 * x[(j * 7919) % n] + 1.0
big = torch.arange(m // 2, m // 2 + extra + 1, dtype=torch.int64, device="cuda")
want_big = 2.0 * float(x[(big * 7919) % n].sum().item()) + 1.0
print(f"m {m} nnz {nnz}", flush=True)
for strat, knobs in [(s, {}) for s in ("adaptive", "line_enhance", "flat", "adaptive_plus", "vector_row", "wf_row", "thread_row", "light", "default")] + [("adaptive", {"guard_full": 1})]:
    lib.spmv_acc_reset_tunables()
    for k, val in knobs.items():
        lib.spmv_acc_set_tunable(k.encode(), val)
    y = torch.ones(m, dtype=torch.float64, device="cuda")
    t0 = time.time()
    try:
        spmv_acc_amd.csr_spmv(2.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        torch.cuda.synchronize()
    except Exception as ex:  # noqa: BLE001
        print(f"  {strat:14s} {knobs} FAILED: {ex}", flush=True)
        lib.spmv_acc_clear_error()
        continue
    dt = time.time() - t0
    bad = 0
    for a in range(0, m, 1 << 28):
        b = min(m, a + (1 << 28))
        rows = torch.arange(a, b, dtype=torch.int64, device="cuda")
        ok = y[a:b] == expected(rows)
        if a <= m // 2 < b:
            ok[m // 2 - a] = True
        bad += int((~ok).sum().item())
        del rows, ok
    big_ok = abs(float(y[m // 2].item()) - want_big) <= 1e-9 * abs(want_big) + 1e-9
    print(f"  {strat:14s} {str(knobs):20s} first call {dt:6.2f} s  wrong rows {bad}  long row {'ok' if big_ok else 'WRONG'}", flush=True)
    spmv_acc_amd.release_plans(rp)
    del y
