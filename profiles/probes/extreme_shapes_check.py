#!/usr/bin/env python3
"""Round 4: the suite's full-size checks (tests/test_gpu_configs.py::_full_size_checks: device reference, alpha / beta, row sums, linearity, a row
prefix against the CPU oracle) on shapes at the edges of what sizes select inside the library -- found worth doing after R-MAT 26 showed a fault that
only x >= 496 MB reaches (profiles/r04_rmat26_check.txt):
  giant_row     one row of 300 M non-zeros among 1,000 short ones, n = 400 M          (long-row slicing, cut runs, merges)
  wide_x        n = 600 M columns (x = 4.8 GB: past the 4 GB the raw-buffer gather hints stop at), 2 M power-law rows
  one_row       m = 1, a dense row of 100 M non-zeros
  diagonal      200 M rows of one non-zero each                                         (rows past 2^27, nothing to balance)
  fat_rows      1,000 rows of 1 M non-zeros each
  mostly_empty  100 M rows, 99 % of them empty, the rest 40 non-zeros
  near_max_fem   (round 5) ~30 M rows of 60 ... 82 non-zeros, nnz = 2^31 - 65537 exactly: the largest count the library accepts (plan.cpp refuses
                 nnz > INT_MAX - 65536) -- every int32 non-zero index at its upper end (2^30 < nnz had never been run: R-MAT 26 is 1.07 B)
  near_max_short (round 5) ~430 M rows of 0 ... 10 non-zeros, the same nnz: rows past 2^28 with non-zero indices past 2^30
under every strategy family, the forced slab passes and the flat tile kernel alone.
    python profiles/probes/extreme_shapes_check.py [shape ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import torch
import spmv_acc_amd
import oracle_lib
import test_gpu_configs as T

oracle_lib.lib()
lib = spmv_acc_amd.load_library()
gen = torch.Generator(device="cuda")
gen.manual_seed(0xE4)


def from_lens(lens, n, skew=False):
    """CSR with the given row lengths; columns uniform (or skewed towards 0) over [0, n), ascending within each row (duplicates allowed)."""
    m = lens.numel()
    rowptr = torch.zeros(m + 1, dtype=torch.int64, device="cuda")
    torch.cumsum(lens, 0, out=rowptr[1:])
    nnz = int(rowptr[-1].item())
    assert nnz <= 2**31 - 65537
    ci = torch.empty(nnz, dtype=torch.int32, device="cuda")
    step = 1 << 27
    r0 = 0
    while r0 < m:  # whole rows per chunk, at most ~step non-zeros (one row may be longer)
        r1 = int(torch.searchsorted(rowptr, rowptr[r0] + step, right=True).item()) - 1
        r1 = min(m, max(r1, r0 + 1))
        s, e = int(rowptr[r0].item()), int(rowptr[r1].item())
        if e > s:
            u = torch.rand(e - s, generator=gen, device="cuda", dtype=torch.float64)
            if skew:
                u = u * u * u
            col = (u * n).long().clamp_(max=n - 1)
            rows = torch.repeat_interleave(torch.arange(r0, r1, device="cuda"), lens[r0:r1], output_size=e - s)
            key, _ = torch.sort((rows - r0) * n + col)
            ci[s:e] = (key % n).to(torch.int32)
            del u, col, rows, key
        r0 = r1
    v = torch.rand(nnz, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    return m, n, nnz, rowptr.to(torch.int32), ci, v


def shape(name):
    i64 = dict(dtype=torch.int64, device="cuda")
    if name == "giant_row":
        lens = torch.randint(1, 40, (1001,), generator=gen, **i64)
        lens[500] = 300_000_000
        return from_lens(lens, 400_000_000)
    if name == "wide_x":
        u = torch.rand(2_000_000, generator=gen, device="cuda", dtype=torch.float64)
        lens = (4.0 / (u + 1e-4) ** 0.6).long().clamp_(max=200_000)
        return from_lens(lens, 600_000_000, skew=True)
    if name == "one_row":
        return from_lens(torch.tensor([100_000_000], **i64), 100_000_000)
    if name == "diagonal":
        m = 200_000_000
        rp = torch.arange(m + 1, device="cuda", dtype=torch.int32)
        return m, m, m, rp, torch.arange(m, device="cuda", dtype=torch.int32), torch.rand(m, generator=gen, device="cuda", dtype=torch.float64) * 2 - 1
    if name == "fat_rows":
        return from_lens(torch.full((1000,), 1_000_000, **i64), 50_000_000)
    if name == "mostly_empty":
        lens = torch.where(torch.rand(100_000_000, generator=gen, device="cuda") < 0.01, 40, 0).long()
        return from_lens(lens, 100_000_000, skew=True)
    if name in ("near_max_fem", "near_max_short"):
        want = 2**31 - 65537
        if name == "near_max_fem":
            lens = torch.randint(60, 83, (31_000_000,), generator=gen, **i64)
        else:
            lens = torch.randint(0, 11, (440_000_000,), generator=gen, **i64)
        # keep the rows up to the one that crosses the largest accepted count, and shorten that one to land on it exactly
        cs = torch.cumsum(lens, 0)
        k = int(torch.searchsorted(cs, torch.tensor([want], **i64)).item())
        lens = lens[: k + 1].clone()
        lens[k] -= int(cs[k].item()) - want
        del cs
        return from_lens(lens, lens.numel(), skew=False)
    raise SystemExit(f"unknown shape {name}")


names = sys.argv[1:] or ["giant_row", "wide_x", "one_row", "diagonal", "fat_rows", "mostly_empty"]
for name in names:
    A = shape(name)
    m, n, nnz, rp, ci, v = A
    lens = rp[1:] - rp[:-1]
    print(f"== {name}: m {m} n {n} nnz {nnz}, longest row {int(lens.max().item())}", flush=True)
    del lens
    x = torch.rand(n, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, device="cuda", dtype=torch.float64)
    for tag, knobs, strats in (("automatic", {}, ("adaptive", "flat", "line_enhance", "adaptive_plus", "vector_row", "wf_row", "light")),
                               ("slab passes forced (16)", {"slab_segments": 16}, ("line_enhance",)),
                               ("slab passes forced (3), every row cut", {"slab_segments": 3, "slab_whole_below": 0}, ("adaptive",)),
                               ("flat tile kernel alone", {"flat_rowblock": 0, "slab_segments": 0}, ("flat",))):
        lib.spmv_acc_reset_tunables()
        for k, val in knobs.items():
            assert lib.spmv_acc_set_tunable(k.encode(), val) == 0, k
        print(f"-- {tag}", flush=True)
        T._full_size_checks(torch, oracle_lib, A, strats, seed=0xE4E4)
        for strat in strats:
            y = y0.clone()
            ms = float(np.median(spmv_acc_amd.time_spmv(strat, 4, 1.0, 1.0, m, n, nnz, rp, ci, v, x, y, y0=y0)))
            info = spmv_acc_amd.query_plan(rp, m)
            print(f"   {strat:14s} {ms * 1e3:10.1f} us   slab_passes {info['slab_passes']}  plus_blocks {info['plus_blocks']}", flush=True)
        spmv_acc_amd.release_plans(rp)
    del A, rp, ci, v, x, y0
    torch.cuda.empty_cache()
lib.spmv_acc_reset_tunables()
print("all checks passed", flush=True)
