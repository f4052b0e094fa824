#!/usr/bin/env python3
"""In-process A/B of the 16-bit column encoding: the same device arrays, tunable col16 = 0 / 1 alternating over `rounds` rounds with the plan released
in between (so every round re-builds the encoding: its allocation's placement is part of what is measured).  Per round and mode: the per-launch
(y reset) median and the back-to-back mean.  usage: tools/ab_col16.py <strategy> <workload,...> [rounds] [extra tunables k=v,...]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

strat, names = sys.argv[1], sys.argv[2].split(",")
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
knobs = [kv.split("=") for kv in (sys.argv[4].split(",") if len(sys.argv) > 4 and sys.argv[4] else [])]
lib = spmv_acc_amd.load_library()
for name in names:
    if name == "banded":
        m, n = 32_000_000, 256_000_000
        rp, ci, v = synth.banded_torch(m, first_row=3 * m, total_rows=n, device="cuda")
        nnz = int(rp[-1].item())
    else:
        m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    y = y0.clone()
    balg = synth.algorithmic_bytes(m, n, nnz)
    res = {0: {"reset": [], "b2b": [], "kernel": []}, 1: {"reset": [], "b2b": [], "kernel": []}}
    used = set()
    for rnd in range(rounds):
        for mode in ((0, 1) if rnd % 2 == 0 else (1, 0)):
            lib.spmv_acc_reset_tunables()
            for k, val in knobs:
                assert lib.spmv_acc_set_tunable(k.encode(), int(val)) == 0
            assert lib.spmv_acc_set_tunable(b"col16", mode) == 0
            spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat)
            ev, kn, _ = spmv_acc_amd.time_spmv_kernels(strat, 30, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)
            iters = 200 if nnz < 20_000_000 else 60
            b2b = spmv_acc_amd.time_spmv_total(strat, iters, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / iters
            info = spmv_acc_amd.query_plan(rp, m)
            used.add((mode, info["col16"], info["stream_policy"], info["last_kernel"]))
            res[mode]["reset"].append(float(np.median(ev)) * 1e3)
            res[mode]["kernel"].append(float(np.median(kn)) * 1e3)
            res[mode]["b2b"].append(b2b * 1e3)
            y.copy_(y0)
            spmv_acc_amd.release_plans(rp)
    line = f"{name:18s} {strat:12s}"
    for mode in (0, 1):
        line += f" | col16={mode}: pair " + " ".join(f"{t:.2f}" for t in res[mode]["reset"]) + "  kernel " + " ".join(f"{t:.2f}" for t in res[mode]["kernel"]) + "  b2b " + " ".join(f"{t:.2f}" for t in res[mode]["b2b"])
    med = {mode: {k: float(np.median(res[mode][k])) for k in res[mode]} for mode in (0, 1)}
    line += " | enc/colindex pair %.4f kernel %.4f b2b %.4f" % tuple(med[1][k] / med[0][k] for k in ("reset", "kernel", "b2b"))
    line += " | frac(kernel) %.3f -> %.3f" % (balg / med[0]["kernel"] / 8e6, balg / med[1]["kernel"] / 8e6)
    print(line + " | " + str(sorted(used)), flush=True)
lib.spmv_acc_reset_tunables()
