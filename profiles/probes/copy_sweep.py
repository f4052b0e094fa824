#!/usr/bin/env python3
"""Streaming-copy ceiling vs footprint and cache policy (non-temporal vs default loads/stores)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, spmv_acc_amd
lib = spmv_acc_amd.load_library()
for mb in (16, 64, 128, 512, 2048, 8192):
    n = mb * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    out = []
    for nt in (1, 0):
        lib.spmv_acc_set_tunable(b"copy_nt", nt)
        out.append(spmv_acc_amd.copy_ceiling_gbs(b, a, reps=5))
    print(f"copy {mb:5d} MiB src + {mb:5d} MiB dst: nt {out[0]:8.1f} GB/s   default policy {out[1]:8.1f} GB/s")
    del a, b
    torch.cuda.empty_cache()
lib.spmv_acc_reset_tunables()
