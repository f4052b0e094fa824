#!/usr/bin/env python3
"""could the whole-row pass of the slab passes (rows below slab_whole_below = 32 non-zeros, each ONE run) go through
stage_products' 16-B path over maximal spans of CONSECUTIVE short rows?  The numbers that decide it, on R-MAT scale 25 (BASELINE configs[3]):
how long those spans are, how many gathers the pass makes, and the gather rate its measured time corresponds to -- against the card's random-sector
ceiling (profiles/r01_gather_microbench.txt: 54 G/s from a 256 MB table)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from spmv_acc_amd import synth

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
pass_us = float(sys.argv[2]) if len(sys.argv) > 2 else 980.0  # segment_tile_kernel<hinted>, profiles/r04_final_rmat_rmat25_line_enhance_kernel_stats.csv
m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
lens = (rp[1:] - rp[:-1]).to(torch.int64)
short = lens < 32
z = torch.zeros(1, dtype=torch.int8, device="cuda")
d = torch.diff(torch.cat([z, short.to(torch.int8), z]))
starts = torch.nonzero(d == 1).flatten()
ends = torch.nonzero(d == -1).flatten()
cs = torch.cat([torch.zeros(1, dtype=torch.int64, device="cuda"), torch.cumsum(lens, 0)])
span_rows = (ends - starts).double()
span_nnz = (cs[ends] - cs[starts]).double()
short_nnz = float(lens[short].sum().item())
print(f"R-MAT {scale}: {m} rows, {nnz} non-zeros; rows below 32 non-zeros: {float(short.double().mean()):.4f} of the rows, {short_nnz / nnz:.4f} of the non-zeros "
      f"({short_nnz / 1e6:.1f} M gathers in the whole-row pass, {float((lens[short] > 0).sum().item()) / 1e6:.1f} M non-empty runs)")
print(f"maximal spans of consecutive short rows: {len(span_rows)} spans, mean {float(span_rows.mean()):.1f} rows / {float(span_nnz.mean()):.1f} non-zeros, "
      f"median {float(span_nnz.median()):.0f} non-zeros; share of the pass's non-zeros in spans of >= 64 / 256 / 2048 non-zeros: "
      + " / ".join(f"{float(span_nnz[span_nnz >= k].sum().item()) / short_nnz:.3f}" for k in (64, 256, 2048)))
print(f"at {pass_us:.0f} us per pass: {short_nnz / pass_us / 1e3:.1f} G gathers/s "
      f"(random-sector ceiling of a {n * 8 / 2**20:.0f} MB x: 54 G/s, profiles/r01_gather_microbench.txt) and "
      f"{(12 * short_nnz + 28 * float((lens[short] > 0).sum().item())) / pass_us / 1e6:.2f} TB/s of stream + run list + y bytes")
