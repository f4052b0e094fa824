#!/usr/bin/env python3
"""In-process A/B of TWO BUILDS of the library (both loaded with ctypes, each with its own plan cache), interleaved round by round on
the same device arrays:
    make -C spmv_acc_amd/csrc OBJ_DIR=build_exp OUT_DIR=../lib_exp EXTRA=-D...     # the experimental build
    python tools/ab_two_libs.py <strategy> <workload,...> [tunables e.g. deterministic=1]
Prints, per workload, the per-launch (y reset) median and the back-to-back mean of both builds over 4 alternating rounds."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

strat, names = sys.argv[1], sys.argv[2].split(",")
knobs = [kv.split("=") for kv in (sys.argv[3].split(",") if len(sys.argv) > 3 and sys.argv[3] else [])]
def raw(path):  # only the entries this script calls, so that an OLDER build of the library (fewer symbols) can stand on either side
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


spmv_acc_amd.load_library()  # (torch's HIP runtime first, see spmv_acc_amd.load_library)
libs = {"shipped": raw(spmv_acc_amd.LIB_PATH), "exp": raw(os.environ.get("AB_EXP_LIB") or os.path.join(ROOT, "spmv_acc_amd", "lib_exp", "libspmv_acc.so"))}
sid = spmv_acc_amd.strategy_id(strat)
for name in names:
    if name == "banded":  # BASELINE configs[4]: rank 3's 32 M-row shard
        m, n = 32_000_000, 256_000_000
        rp, ci, v = synth.banded_torch(m, first_row=3 * m, total_rows=n, device="cuda")
        nnz = int(rp[-1].item())
    else:
        m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    res = {k: {"reset": [], "b2b": []} for k in libs}
    iters = 200 if nnz < 20_000_000 else 60
    for rnd in range(4):
        for key, lib in libs.items():
            for k, val in knobs:
                assert lib.spmv_acc_set_tunable(k.encode(), int(val)) == 0
            if key == "exp":  # AB_EXP_KNOBS=k=v,...: tunables for the experimental build only (a build whose constants ask for other settings)
                for kv in filter(None, os.environ.get("AB_EXP_KNOBS", "").split(",")):
                    k, val = kv.split("=")
                    assert lib.spmv_acc_set_tunable(k.encode(), int(val)) == 0
            args = (m, n, nnz, None, rp.data_ptr(), ci.data_ptr(), v.data_ptr(), x.data_ptr(), y.data_ptr())
            for _ in range(5):
                lib.spmv_acc_csr_spmv_strategy(sid, 0, 1.0, 1.0, *args)
            torch.cuda.synchronize()
            out = (ctypes.c_float * 30)()
            assert lib.spmv_acc_time_spmv(sid, 30, 1.0, 1.0, *args, y0.data_ptr(), ctypes.cast(out, ctypes.c_void_p)) == 0
            res[key]["reset"].append(float(np.median(list(out))) * 1e3)
            tot = ctypes.c_float(0)
            assert lib.spmv_acc_time_spmv_total(sid, iters, 1.0, 1.0, *args, ctypes.addressof(tot)) == 0
            res[key]["b2b"].append(tot.value / iters * 1e3)
            y.copy_(y0)
    line = f"{name:18s} {strat:13s}"
    for key in libs:
        line += f" | {key}: reset " + " ".join(f"{t:.2f}" for t in res[key]["reset"]) + "  b2b " + " ".join(f"{t:.2f}" for t in res[key]["b2b"])
    a, b = np.median(res["shipped"]["reset"]), np.median(res["exp"]["reset"])
    a2, b2 = np.median(res["shipped"]["b2b"]), np.median(res["exp"]["b2b"])
    print(line + f" | shipped/exp reset {a / b:.4f} b2b {a2 / b2:.4f}", flush=True)
    for lib in libs.values():
        lib.spmv_acc_release_plans(None)
