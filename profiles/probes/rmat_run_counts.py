#!/usr/bin/env python3
"""Round 6: why more column slabs do not help the slab-major copy on R-MAT 25 -- the number of (row, slab) runs per slab count.
Every run is one y read-modify-write (16 B) + a compact-row entry (row id 4 B + row pointer 4 B) on top of the matrix' own 12 B per non-zero."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from spmv_acc_amd import synth
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
del v
lens = (rp[1:] - rp[:-1])
rows = torch.repeat_interleave(torch.arange(m, device="cuda", dtype=torch.int32), lens.long())
first = torch.ones(nnz, dtype=torch.bool, device="cuda")
first[1:] = rows[1:] != rows[:-1]
print(f"R-MAT {scale}: m {m} nnz {nnz} matrix stream {12 * nnz / 1e9:.2f} GB")
for S in (1, 4, 8, 12, 16, 32, 64, 128):
    width = (n + S - 1) // S
    slab = (ci // width).to(torch.int16)
    new_run = first.clone()
    new_run[1:] |= slab[1:] != slab[:-1]
    runs = int(new_run.sum().item())
    print(f"S {S:3d}: x window {8 * width / 1e6:7.1f} MB  runs {runs / 1e6:7.1f} M ({runs / m:.2f} per row, {nnz / runs:.1f} non-zeros per run)  "
          f"run overhead {24 * runs / 1e9:5.2f} GB = {24 * runs / (12 * nnz):.2f} x the matrix stream", flush=True)
