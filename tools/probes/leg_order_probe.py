#!/usr/bin/env python3
"""bench.py's sweep legs run `flat` first, then `adaptive`; on balanced matrices both end up in the same row-block kernel, yet the flat leg reads 2-6 %
slower in some runs.  Order, or strategy?  The two legs alternated three times on fresh plans, same arrays, in one process.
    python tools/probes/leg_order_probe.py [name ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import spmv_acc_amd
from spmv_acc_amd import synth
import bench

device = torch.device("cuda", 0)
for name in sys.argv[1:] or ["af_shell10", "Bump_2911", "Cube_Coup_dt6"]:
    A = synth.sweep_standin_torch(name, device=device)
    x, y0 = bench._leg_vectors(torch, device, A[0], A[1])
    out = []
    for rep in range(3):
        for strat in ("flat", "adaptive"):
            r = bench.timed_leg(torch, strat, A, x, y0, 60)
            out.append(f"{strat} {r['us']:.1f} {r.get('plan')}")
            spmv_acc_amd.release_plans(A[3])
    print(f"{name}: " + " | ".join(out), flush=True)
    del A, x, y0
    torch.cuda.empty_cache()
