#!/usr/bin/env python3
"""The same stand-in in torch's caching allocator (as bench.py holds it) and in one hipMalloc per array (spmv_acc_stage_csr: how the reference's harness and
any C / C++ caller hold a matrix, cli/utils.hpp:104-114), in a process that also holds the headline matrix.  adaptive / flat, per-launch protocol, us."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, spmv_acc_amd
from spmv_acc_amd import synth
lib = spmv_acc_amd.load_library()
head = synth.hardesty3_like_torch(device="cuda")
def measure(m, n, nnz, rp, ci, v, x, y, y0, strat):
    for _ in range(10):
        spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    us = float(np.median(spmv_acc_amd.time_spmv(strat, 60, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0))) * 1e3
    spmv_acc_amd.release_plans(rp)
    return us
for name in (sys.argv[1:] or ["Bump_2911", "Cube_Coup_dt6", "vas_stokes_2M", "af_shell10"]):
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    y = y0.clone()
    a = {s: measure(m, n, nnz, rp, ci, v, x, y, y0, s) for s in ("flat", "adaptive")}
    h = [t.cpu().numpy() for t in (rp, ci, v, x, y0)]
    del rp, ci, v, x, y
    torch.cuda.empty_cache()
    ptrs = [ctypes.c_void_p() for _ in range(5)]
    rc = lib.spmv_acc_stage_csr(m, n, nnz, *(arr.ctypes.data for arr in h), *(ctypes.byref(p) for p in ptrs))
    assert rc == 0, lib.spmv_acc_last_error_string()
    torch.cuda.synchronize()
    d = [p.value for p in ptrs]
    b = {s: measure(m, n, nnz, d[0], d[1], d[2], d[3], d[4], y0, s) for s in ("flat", "adaptive")}
    for p in d:
        lib.spmv_acc_free_device(ctypes.c_void_p(p))
    print(f"{name}: torch allocator flat {a['flat']:.1f} adaptive {a['adaptive']:.1f} | one hipMalloc per array flat {b['flat']:.1f} adaptive {b['adaptive']:.1f}", flush=True)
    del y0
    torch.cuda.empty_cache()
