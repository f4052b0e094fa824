#!/usr/bin/env python3
"""Streaming-copy ceiling vs footprint (is the SpMV slow-down on 4 GB matrices a property of the card?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, spmv_acc_amd
for mb in (8, 16, 32, 64, 128, 512, 1024, 2048, 4096, 8192):
    n = mb * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
    b = torch.empty_like(a)
    print(f"copy {mb:5d} MiB src + {mb:5d} MiB dst: {spmv_acc_amd.copy_ceiling_gbs(b, a, reps=5):8.1f} GB/s")
    del a, b
    torch.cuda.empty_cache()
