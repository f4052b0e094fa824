#!/usr/bin/env python3
"""Writes a boneS10-sized/10 bin2 file and runs spmv-cli in CLI and --benchmark mode (reference protocols + CSV schema)."""
import os, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from spmv_acc_amd import synth
m, n, nnz, rp, ci, v = synth.large_set_like_torch("boneS10", device="cuda", scale=0.25)
d = tempfile.mkdtemp()
p = os.path.join(d, "boneS10_like_quarter.bin2")
with open(p, "wb") as f:
    f.write(struct.pack("<6i", 0x20211015, 2, 3, m, n, nnz))
    rp.cpu().numpy().astype("<i4").tofile(f); ci.cpu().numpy().astype("<i4").tofile(f); v.cpu().numpy().astype("<f8").tofile(f)
cli = os.path.join(ROOT, "spmv_acc_amd", "bin", "spmv-cli")
for args in (["-f", "bin2"], ["-f", "bin2", "--benchmark"]):
    r = subprocess.run([cli, p] + args, capture_output=True, text=True)
    print("$ spmv-cli", os.path.basename(p), " ".join(args)); print(r.stdout.strip()); print(r.stderr.strip()[:300])
