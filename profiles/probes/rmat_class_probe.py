#!/usr/bin/env python3
"""Round 4: would per-CLASS slab counts pay on R-MAT 25?  The slab passes cut every row (of >= 32 non-zeros) into the same S = 8 column slabs; the
line re-fetch argument says long rows want more slabs and medium rows fewer.  This probe splits the matrix into row classes by
length -- each class as a matrix of its own (the other rows emptied, same columns) -- and times every class under the one-kernel path and under forced
S = 2 / 4 / 8 / 12 / 15 passes (slab_whole_below = 0): the sum of the per-class minima is what a multi-class form could reach at best.
    python profiles/probes/rmat_class_probe.py [scale=25]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 25
m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device="cuda", seed=0xC4)
lib = spmv_acc_amd.load_library()
lens = (rp[1:] - rp[:-1]).long()
rows_of = torch.repeat_interleave(torch.arange(m, device="cuda"), lens, output_size=nnz)
x = torch.rand(n, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, device="cuda", dtype=torch.float64)
classes = [(0, 32), (32, 128), (128, 512), (512, 2048), (2048, 1 << 30)]
if os.environ.get("CLASS_ONLY"):  # e.g. CLASS_ONLY=2048: that class alone (for a rocprofv3 run), S = 8 only
    lo_only = int(os.environ["CLASS_ONLY"])
    classes = [c for c in classes if c[0] == lo_only]
variants = [("one kernel", {"slab_segments": 0})] + [(f"S={S}", {"slab_segments": S, "slab_whole_below": 0}) for S in (2, 4, 8, 12, 15)]
if os.environ.get("CLASS_ONLY"):
    variants = [("S=8", {"slab_segments": 8, "slab_whole_below": 0})]
if len(sys.argv) > 2:  # a second strategy beside line_enhance, un-slabbed (e.g. wf_row: one wavefront per row, 16-B loads, no tile machinery)
    variants.append((sys.argv[2] + " (no slabs)", {"slab_segments": 0, "__strategy": sys.argv[2]}))
best_sum = 0.0
print(f"R-MAT {scale}: m {m} nnz {nnz}", flush=True)
for lo, hi in classes:
    keep_row = (lens >= lo) & (lens < hi)
    keep = keep_row[rows_of]
    sub_lens = torch.where(keep_row, lens, torch.zeros_like(lens))
    srp = torch.zeros(m + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(sub_lens, 0, out=srp[1:])
    srp = srp.to(torch.int32)
    sci, sv = ci[keep].contiguous(), v[keep].contiguous()
    snnz = int(srp[-1].item())
    line = f"rows of [{lo}, {hi if hi < 1 << 29 else 'inf'}) non-zeros: {int(keep_row.sum().item())} rows, {snnz} nnz ({100.0 * snnz / nnz:.1f} %):"
    times = {}
    for name, knobs in variants:
        lib.spmv_acc_reset_tunables()
        strat = knobs.get("__strategy", "line_enhance")
        for k, val in knobs.items():
            if not k.startswith("__"):
                lib.spmv_acc_set_tunable(k.encode(), val)
        y = y0.clone()
        spmv_acc_amd.prepare(m, n, snnz, srp, sci, sv, x, strategy=strat)
        for _ in range(2):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, snnz, srp, sci, sv, x, y, strategy=strat)
        torch.cuda.synchronize()
        ms = float(np.median(spmv_acc_amd.time_spmv(strat, 6, 1.0, 1.0, m, n, snnz, srp, sci, sv, x, y, y0=y0)))
        times[name] = ms
        spmv_acc_amd.release_plans(srp)
    best = min(times, key=times.get)
    best_sum += times[best]
    print(line + "  " + "  ".join(f"{k} {t * 1e3:7.1f} us" for k, t in times.items()) + f"   -> {best}", flush=True)
    del srp, sci, sv, keep, keep_row, sub_lens
    torch.cuda.empty_cache()
lib.spmv_acc_reset_tunables()
print(f"sum of the per-class minima: {best_sum * 1e3:.1f} us (every class pays its own pass over y here: an upper bound of what a multi-class form would take)")
