#!/usr/bin/env python3
"""Round 4: the headline matrix's far gathers (10 % of the non-zeros, uniformly random columns) are what keeps it at 0.56.  Round 2 tried issuing them
NON-TEMPORAL (20-25 % slower: x lives in the Infinity Cache, nt forfeits the hits).  Not tried then: the scope bits -- sc0 / sc1 bypass the CU's
vector L1 (and, at sc1, make the L2 treat the line as coherent) without the streaming hint.  Experimental builds mark cold = far from both stream
neighbours (-DSPMV_ACC_HINT_BY_POSITION) and issue cold gathers with aux 1 / 2 / 16 / 17 (-DSPMV_ACC_COLD_AUX=...), gather_hint = 1 uses the marks:
    for a in 1 2 16 17; do make -C spmv_acc_amd/csrc -j8 OBJ_DIR=build_aux$a OUT_DIR=../lib_aux$a EXTRA="-DSPMV_ACC_HINT_BY_POSITION -DSPMV_ACC_COLD_AUX=$a"; done
    python profiles/probes/far_gather_policy_ab.py [workload ...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth


def raw(path):
    lib = ctypes.CDLL(path)
    vp, ci, cd = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    lib.spmv_acc_csr_spmv_strategy.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_csr_spmv_strategy.restype = None
    lib.spmv_acc_time_spmv.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_time_spmv_total.argtypes = [ci, ci, cd, cd, ci, ci, ci, vp, vp, vp, vp, vp, vp, vp]
    lib.spmv_acc_set_tunable.argtypes = [ctypes.c_char_p, ci]
    lib.spmv_acc_release_plans.argtypes = [vp]
    lib.spmv_acc_release_plans.restype = None
    return lib


spmv_acc_amd.load_library()
libs = {"shipped": (raw(spmv_acc_amd.LIB_PATH), {})}
for a in (1, 2, 16, 17):
    p = os.path.join(ROOT, "spmv_acc_amd", f"lib_aux{a}", "libspmv_acc.so")
    if os.path.exists(p):
        libs[f"far gathers aux {a}"] = (raw(p), {"gather_hint": 1})
sid = spmv_acc_amd.strategy_id("adaptive")
for name in sys.argv[1:] or ["Hardesty3"]:
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    ref = None
    res = {k: {"reset": [], "b2b": []} for k in libs}
    for rnd in range(3):
        for key, (lib, knobs) in libs.items():
            for k, val in knobs.items():
                assert lib.spmv_acc_set_tunable(k.encode(), val) == 0
            args = (m, n, nnz, None, rp.data_ptr(), ci.data_ptr(), v.data_ptr(), x.data_ptr(), y.data_ptr())
            y.copy_(y0)
            lib.spmv_acc_csr_spmv_strategy(sid, 0, 1.0, 1.0, *args)
            torch.cuda.synchronize()
            if ref is None:
                ref = y.clone()
            err = float((y - ref).abs().max().item())
            assert err < 1e-9, (key, err)
            for _ in range(5):
                lib.spmv_acc_csr_spmv_strategy(sid, 0, 1.0, 1.0, *args)
            torch.cuda.synchronize()
            out = (ctypes.c_float * 30)()
            assert lib.spmv_acc_time_spmv(sid, 30, 1.0, 1.0, *args, y0.data_ptr(), ctypes.cast(out, ctypes.c_void_p)) == 0
            res[key]["reset"].append(float(np.median(list(out))) * 1e3)
            tot = ctypes.c_float(0)
            assert lib.spmv_acc_time_spmv_total(sid, 60, 1.0, 1.0, *args, ctypes.addressof(tot)) == 0
            res[key]["b2b"].append(tot.value / 60 * 1e3)
    for key in libs:
        print(f"{name:12s} {key:22s} per launch " + " ".join(f"{t:7.2f}" for t in res[key]["reset"]) + "   back to back " + " ".join(f"{t:7.2f}" for t in res[key]["b2b"]), flush=True)
    for lib, _ in libs.values():
        lib.spmv_acc_release_plans(None)
