#!/usr/bin/env python3
"""Does the ranking of the stream cache policies depend on WHERE y lives?  Same matrix, same plan, pinned policy, several y
buffers (fresh allocations separated by pads of different sizes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

lib = spmv_acc_amd.load_library()
m, n, nnz = synth.LARGE_SET["Hardesty3"]
far = float(sys.argv[1]) if len(sys.argv) > 1 else 0.0
rp, ci, v = synth.structured_csr_torch(m, n, nnz, 0xC2, device="cuda", far_fraction=far)
x = torch.rand(n, device="cuda", dtype=torch.float64)
pads, ys = [], []
for k, padmb in enumerate((0, 1, 3, 7, 16, 33, 64, 100)):
    pads.append(torch.empty(padmb << 20, dtype=torch.uint8, device="cuda"))
    ys.append(torch.zeros(m, device="cuda", dtype=torch.float64))
torch.cuda.synchronize()
for pol in (0, 1, 3):
    lib.spmv_acc_reset_tunables()
    lib.spmv_acc_set_tunable(b"stream_plain", pol)
    row = []
    for y in ys:
        for _ in range(3):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy="line_enhance")
        t = spmv_acc_amd.time_spmv_total("line_enhance", 60, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / 60
        row.append(f"{t * 1e3:6.1f}")
    print(f"far {far} policy {pol}: us per y buffer (addr mod 2 MiB in KiB: {[ (y.data_ptr() % (2 << 20)) >> 10 for y in ys]}):", " ".join(row), flush=True)
lib.spmv_acc_reset_tunables()
