#!/usr/bin/env python3
"""placement_probe.py with knobs: for each placement of one stand-in, back-to-back time under several pinned variants -- does any of
them recover what an unlucky placement loses?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import spmv_acc_amd
from spmv_acc_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "Cube_Coup_dt6"
strat = "line_enhance"
lib = spmv_acc_amd.load_library()
variants = ["stream_plain=1", "stream_plain=3", "stream_plain=2", "stream_plain=0", "stream_plain=1,zigzag=0", "stream_plain=1,xcd_chunk=0", "stream_plain=1,xcd_chunk=64",
            "stream_plain=1,xcd_remap=1", "stream_plain=1,rowblock_target=1500"]
keep = []
print("variants:", variants)
for k, padmb in enumerate((0, 1, 64, 200, 0, 0, 0)):
    if padmb:
        keep.append(torch.empty(padmb << 20, dtype=torch.uint8, device="cuda"))
    if k >= 4:
        keep.clear()
        torch.cuda.empty_cache()
    m, n, nnz, rp, ci, v = synth.sweep_standin_torch(name)
    x = torch.rand(n, device="cuda", dtype=torch.float64)
    y = torch.rand(m, device="cuda", dtype=torch.float64)
    row = []
    for var in variants:
        lib.spmv_acc_reset_tunables()
        lib.spmv_acc_set_tunable(b"deterministic", 1)
        for kv in var.split(","):
            a, b = kv.split("=")
            lib.spmv_acc_set_tunable(a.encode(), int(b))
        spmv_acc_amd.release_plans(rp)
        for _ in range(6):
            spmv_acc_amd.csr_spmv(1.0, 1.0, m, n, nnz, rp, ci, v, x, y, strategy=strat)
        row.append(spmv_acc_amd.time_spmv_total(strat, 100, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y) / 100 * 1e3)
    print(f"{name} pad {padmb:4d}: " + " ".join(f"{t:7.2f}" for t in row) + f"   v-ci offset {(v.data_ptr() - ci.data_ptr()) / (1 << 20):10.2f} MiB", flush=True)
    spmv_acc_amd.release_plans(rp)
    del rp, ci, v, x, y
lib.spmv_acc_reset_tunables()
