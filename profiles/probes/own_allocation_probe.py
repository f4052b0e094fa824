#!/usr/bin/env python3
"""Does it matter that a generated stand-in's arrays lie wherever the generator's temporaries left them inside the caching allocator's segments?
Each large sweep stand-in as generated, then the SAME arrays cloned into allocations of their own (what a caller that hipMallocs each array holds --
the reference's harness does, cli/utils.hpp:104-114), adaptive and flat, per-launch protocol, in a process that also holds the headline matrix."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, spmv_acc_amd
from spmv_acc_amd import synth
head = synth.hardesty3_like_torch(device="cuda")
def measure(A, strat):
    m, n, nnz, rp, ci, v = A
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    y = y0.clone()
    for _ in range(10):
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    us = float(np.median(spmv_acc_amd.time_spmv(strat, 60, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))) * 1e3
    spmv_acc_amd.release_plans(rp)
    return us
for name in (sys.argv[1:] or ["boneS10", "Bump_2911", "Cube_Coup_dt6", "vas_stokes_2M", "af_shell10", "dielFilterV3real"]):
    A = synth.sweep_standin_torch(name)
    as_made = {s: measure(A, s) for s in ("flat", "adaptive")}
    m, n, nnz, rp, ci, v = A
    torch.cuda.empty_cache()
    B = (m, n, nnz, rp.clone(), ci.clone(), v.clone())
    del A, rp, ci, v
    torch.cuda.empty_cache()
    own = {s: measure(B, s) for s in ("flat", "adaptive")}
    offs = [(t.data_ptr() % (1 << 21)) for t in B[3:]]
    print(f"{name}: as generated flat {as_made['flat']:.1f} adaptive {as_made['adaptive']:.1f} | own allocations flat {own['flat']:.1f} adaptive {own['adaptive']:.1f}  (offsets in 2 MB: {offs})", flush=True)
    del B
    torch.cuda.empty_cache()
