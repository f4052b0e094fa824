#!/usr/bin/env python3
"""Kernel A/B harness: N strategies x M rounds interleaved in ONE process on one device (per-launch
hipEvent times from the library), reporting median / min and the algorithmic-bytes bandwidth.

    python tools/kbench.py --workloads hardesty3,boneS10 --strategies adaptive,flat --rounds 5 --iters 20
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--workloads", default="hardesty3")
    p.add_argument("--strategies", default="adaptive,flat,line_enhance,adaptive_plus,default")
    p.add_argument("--rounds", type=int, default=5)
    p.add_argument("--iters", type=int, default=20)
    p.add_argument("--scale", type=float, default=1.0)
    p.add_argument("--tunables", default="", help="semicolon-separated variants, each 'name=val,name=val' (empty = defaults)")
    p.add_argument("--beta", type=float, default=1.0)
    p.add_argument("--far", type=float, default=None, help="override the far-column fraction of the stand-in")
    a = p.parse_args()
    import torch

    import spmv_acc_amd
    from spmv_acc_amd import synth

    lib = spmv_acc_amd.load_library()
    variants = [v for v in a.tunables.split(";")] if a.tunables else [""]
    for w in a.workloads.split(","):
        if w == "hardesty3":
            if a.far is None:
                m, n, nnz, rp, ci, v = synth.hardesty3_like_torch(device="cuda", scale=a.scale)
            else:
                m, n, nnz = synth.LARGE_SET["Hardesty3"]
                rp, ci, v = synth.structured_csr_torch(m, n, nnz, 0xC2, device="cuda", far_fraction=a.far)
        elif w == "banded":
            rows = int(32_000_000 * a.scale)
            rp, ci, v = synth.banded_torch(rows, device="cuda")
            m = n = rows
            nnz = int(rp[-1].item())
        elif w.startswith("rmat"):
            m, n, nnz, rp, ci, v = synth.rmat_torch(int(w[4:] or 22), device="cuda")
        else:
            m, n, nnz, rp, ci, v = synth.large_set_like_torch(w, device="cuda", scale=a.scale)
        g = torch.Generator(device="cuda")
        g.manual_seed(1)
        x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
        y = torch.rand(m, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
        balg = synth.algorithmic_bytes(m, n, nnz, beta_nonzero=a.beta != 0)
        combos = [(s, t) for s in a.strategies.split(",") for t in variants]
        times = {c: [] for c in combos}
        for r in range(a.rounds):
            for s, t in combos:
                if hasattr(lib, "spmv_acc_set_tunable"):
                    lib.spmv_acc_reset_tunables()
                    for kv in filter(None, t.split(",")):
                        k, val = kv.split("=")
                        assert lib.spmv_acc_set_tunable(k.encode(), int(val)) == 0, kv
                ms = spmv_acc_amd.time_spmv(s, a.iters, 1.0, a.beta, m, n, nnz, rp, ci, v, x, y)
                times[(s, t)] += list(ms[2:])
        info = spmv_acc_amd.query_plan(rp, m) or {}
        print(f"== {w}: m={m} n={n} nnz={nnz} avg={nnz / m:.2f} B_alg={balg / 1e6:.1f} MB  (plan: stream_policy={info.get('stream_policy')})")
        for (s, t), ms in times.items():
            med, mn = float(np.median(ms)), float(np.min(ms))
            print(f"  {s:14s} {t:28s} median {med * 1e3:9.2f} us  min {mn * 1e3:9.2f} us  "
                  f"{balg / med / 1e6:8.1f} GB/s  {2 * nnz / med / 1e6:8.1f} GFLOP/s")
        sys.stdout.flush()
        spmv_acc_amd.release_plans()
        del rp, ci, v, x, y
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
