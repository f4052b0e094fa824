#!/usr/bin/env python3
"""On arrays held in one hipMalloc each (all 2 MB-aligned: the placement a C / C++ caller gets), which engine knobs move the large sweep stand-ins?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, spmv_acc_amd
from spmv_acc_amd import synth
lib = spmv_acc_amd.load_library()
strat = "line_enhance"
variants = ["", "xcd_chunk=0", "xcd_chunk=4", "xcd_chunk=64", "zigzag=0", "stream_plain=0", "stream_plain=1", "stream_plain=3", "cache_ends_mb=0", "rowblock_target=1900", "rowblock_target=1200", "early_y=0", "xcd_remap=1,xcd_chunk=0"]
for name in (sys.argv[1:] or ["af_shell10", "Bump_2911"]):
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    gen = torch.Generator(device="cuda"); gen.manual_seed(1234)
    x = torch.rand(n, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    h = [t.cpu().numpy() for t in (rp, ci, v, x, y0)]
    del rp, ci, v, x
    torch.cuda.empty_cache()
    ptrs = [ctypes.c_void_p() for _ in range(5)]
    assert lib.spmv_acc_stage_csr(m, n, nnz, *(arr.ctypes.data for arr in h), *(ctypes.byref(p) for p in ptrs)) == 0
    torch.cuda.synchronize()
    d = [p.value for p in ptrs]
    res = {}
    for rnd in range(2):
        for var in variants:
            lib.spmv_acc_reset_tunables()
            lib.spmv_acc_set_tunable(b"deterministic", 1)
            for kv in filter(None, var.split(",")):
                k, val = kv.split("=")
                lib.spmv_acc_set_tunable(k.encode(), int(val))
            spmv_acc_amd.release_plans(d[0])
            for _ in range(6):
                spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, d[0], d[1], d[2], d[3], d[4], strategy=strat)
            res.setdefault(var, []).append(float(np.median(spmv_acc_amd.time_spmv(strat, 40, 1.0, 1.0, m, n, nnz, d[0], d[1], d[2], d[3], d[4], y0=y0))) * 1e3)
    base = min(res[""])
    print(f"{name} {strat} (hipMalloc per array): " + " | ".join(f"[{var or 'default'}] {min(t):.1f} ({min(t) / base:.3f})" for var, t in res.items()), flush=True)
    lib.spmv_acc_reset_tunables()
    spmv_acc_amd.release_plans(d[0])
    for p in d:
        lib.spmv_acc_free_device(ctypes.c_void_p(p))
