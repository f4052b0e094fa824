#!/usr/bin/env python3
"""Which thread count / placement makes bench.py's cpu_baseline (the oracle on the host cores) fast AND stable on the GPU box?  The box shows 256
hardware threads and grants a cgroup CPU quota of 16 (profiles/r05_*: `cgroup_cpu_quota`); earlier rounds ran 128 unpinned threads over numpy-allocated
arrays and read 20-40 GFLOP/s from run to run.  The headline matrix through bench.py's own child (`--cpu-baseline-child`) under several settings."""
import json
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from spmv_acc_amd import synth

m, n, nnz, rp, ci, v = synth.hardesty3_like_torch(device="cuda", seed=0xC2)
g = torch.Generator(device="cuda")
g.manual_seed(1234)
x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
y0 = torch.rand(m, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
d = tempfile.mkdtemp(prefix="cpu_probe_", dir="/dev/shm")
try:
    for k, t in (("rp", rp), ("ci", ci), ("v", v), ("x", x), ("y0", y0)):
        np.save(os.path.join(d, f"{k}.npy"), t.cpu().numpy())
    del rp, ci, v, x, y0
    torch.cuda.empty_cache()
    for bind, places, threads in (("close", "cores", 16), ("spread", "cores", 16), ("spread", "cores", 32), ("spread", "cores", 64), ("spread", "cores", 128),
                                  ("close", "threads", 256), ("false", "", 128), ("spread", "cores", 16)):
        env = dict(os.environ, OMP_PROC_BIND=bind, OMP_NUM_THREADS=str(threads), OMP_DYNAMIC="false", SPMV_ACC_CPU_THREADS=str(threads))
        if places:
            env["OMP_PLACES"] = places
        else:
            env.pop("OMP_PLACES", None)
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-child", d, "--cpu-seconds", "6"], env=env, capture_output=True,
                           text=True, timeout=600)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        if not lines:
            print(f"bind {bind} places {places} threads {threads}: failed: {r.stderr[-300:]}")
            continue
        o = json.loads(lines[-1])
        print(f"OMP_PROC_BIND={bind:6s} OMP_PLACES={places or '-':7s} threads {threads:3d}: best {o['value']:7.2f} GFLOP/s, round medians {o['value_median_per_round']}, "
              f"spread {o['spread_of_round_medians']:.2f}, triad {o['stream_triad_gbs']:.0f} GB/s, achieved {o['achieved_gbs']:.0f} GB/s, 1 thread {o['value_1thread']}", flush=True)
finally:
    shutil.rmtree(d, ignore_errors=True)
