#!/usr/bin/env python3
"""Round 4: per-class slab counts on R-MAT (tunables slab_whole_below / slab_short_* / slab_mid_*): time the forced slab passes (slab_segments = 8 for
the longest rows) under a list of class tables, checking every result against the one-kernel path.
    python tools/probes/rmat_slab_classes_sweep.py [scale=25]"""
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
x = torch.rand(n, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, device="cuda", dtype=torch.float64)
print(f"R-MAT {scale}: m {m} nnz {nnz}", flush=True)


def run(knobs, strat="line_enhance"):
    lib.spmv_acc_reset_tunables()
    for k, val in knobs.items():
        assert lib.spmv_acc_set_tunable(k.encode(), val) == 0, k
    y = y0.clone()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
    t1.record(); torch.cuda.synchronize()
    prep = t0.elapsed_time(t1)
    spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    torch.cuda.synchronize()
    ms = float(np.median(spmv_acc_amd.time_spmv(strat, 8, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y.clone(), y0=y0)))
    spmv_acc_amd.release_plans(rp)
    return ms, prep, y


ref_ms, _, ref = run({"slab_segments": 0})
print(f"one kernel: {ref_ms * 1e3:8.1f} us", flush=True)
scale_y = float(ref.abs().max().item())
tables = [
    ("round-4 two-class (whole < 32)", dict(slab_whole_below=32, slab_short_below=0, slab_mid_below=0)),
    ("mid [32,128) x4", dict(slab_whole_below=32, slab_short_below=0, slab_mid_below=128, slab_mid_slabs=4)),
    ("mid [32,256) x4", dict(slab_whole_below=32, slab_short_below=0, slab_mid_below=256, slab_mid_slabs=4)),
    ("mid [32,128) x3", dict(slab_whole_below=32, slab_short_below=0, slab_mid_below=128, slab_mid_slabs=3)),
    ("short [0,32) x2, mid x4", dict(slab_whole_below=0, slab_short_below=32, slab_short_slabs=2, slab_mid_below=128, slab_mid_slabs=4)),
    ("whole < 4, short [4,32) x2, mid x4", dict(slab_whole_below=4, slab_short_below=32, slab_short_slabs=2, slab_mid_below=128, slab_mid_slabs=4)),
    ("whole < 8, short [8,32) x2, mid x4", dict(slab_whole_below=8, slab_short_below=32, slab_short_slabs=2, slab_mid_below=128, slab_mid_slabs=4)),
    ("whole < 16, short [16,32) x2, mid x4", dict(slab_whole_below=16, slab_short_below=32, slab_short_slabs=2, slab_mid_below=128, slab_mid_slabs=4)),
    ("whole < 8, short [8,48) x2, mid [48,192) x4", dict(slab_whole_below=8, slab_short_below=48, slab_short_slabs=2, slab_mid_below=192, slab_mid_slabs=4)),
    ("whole < 8, short [8,32) x3, mid x5", dict(slab_whole_below=8, slab_short_below=32, slab_short_slabs=3, slab_mid_below=128, slab_mid_slabs=5)),
]
if os.environ.get("SWEEP_EXTRA"):
    for spec in os.environ["SWEEP_EXTRA"].split(";"):
        vals = [int(t) for t in spec.split(",")]
        tables.append((spec, dict(slab_whole_below=vals[0], slab_short_below=vals[1], slab_short_slabs=vals[2], slab_mid_below=vals[3], slab_mid_slabs=vals[4])))
for name, knobs in tables:
    for S in (8,) if not os.environ.get("SWEEP_S") else [int(t) for t in os.environ["SWEEP_S"].split(",")]:
        ms, prep, y = run(dict(knobs, slab_segments=S))
        err = float((y - ref).abs().max().item()) / scale_y
        print(f"S={S:2d} {name:46s} {ms * 1e3:8.1f} us   prepare {prep:7.1f} ms   max |dy| / max |y| {err:.2e}", flush=True)
        assert err < 1e-12, err
