"""GPU suite (-m gpu), BASELINE.json configs[2..4] at FULL size through the C ABI.

configs[2]  the 12 stand-ins of the large-data-set sweep (examples/large-data-set-batch.sh:24-52 dims + scircuit / af_shell10),
            strategies flat (the one BASELINE names) and adaptive
configs[3]  R-MAT scale 25, edge factor 16 (~0.53 B non-zeros), line_enhance
configs[4]  one 32 M-row shard of the 256 M-row banded matrix (global column ids), adaptive / flat / line_enhance

The CPU oracle cannot walk these sizes in seconds, so each matrix is checked three ways (the pattern of
tests/test_gpu_parity.py::test_full_size_*):
  * against the oracle (oracle_host_spmv = cli/verification.cpp:56-66) on a row PREFIX of the full-size run,
  * against an independent fp64 evaluation on the device (torch's segmented sum over the products, another summation order),
    with the scaled-error gate and the reference's own verify_y rule (cli/verification.cpp:15-38),
  * through size-independent properties: row sums for x = 1, linearity in x, y = alpha*(A x) + beta*y0.
Tolerances as in test_gpu_parity.py: scaled error <= 1e-12, reference verdict rel 1e-7 / abs 1e-14.
"""
import time

import numpy as np
import pytest

import spmv_acc_amd
from spmv_acc_amd import synth

pytestmark = pytest.mark.gpu

SCALED_TOL = 1e-12


@pytest.fixture(scope="module")
def torch_dev(hiplib):
    import torch

    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return torch


def _vectors(torch, m, n, seed):
    g = torch.Generator(device="cuda")
    g.manual_seed(seed)
    x = torch.rand(n, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, generator=g, device="cuda", dtype=torch.float64) * 2 - 1
    return x, y0


def _spmv(torch, A, strat, alpha, beta, x, y0):
    m, n, nnz, rp, ci, v = A
    y = y0.clone()
    spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    torch.cuda.synchronize()
    return y


_T0 = time.time()


def _say(msg):
    print(f"[configs +{time.time() - _T0:6.1f}s] {msg}", flush=True)  # (run with -s on the GPU box: a long test is then seen to be alive)


def _device_reference(torch, A, x):
    """(A x, |A| |x|) in fp64 on the device, independent of the library: torch's segmented sum over the products (one
    sequential sum per row: no atomics, so a hub row of millions of non-zeros is not a contention hot spot), in chunks of
    rows so the temporaries of a 0.5 B-non-zero matrix stay small."""
    m, n, nnz, rp, ci, v = A
    ax = torch.zeros(m, dtype=torch.float64, device="cuda")
    mag = torch.zeros(m, dtype=torch.float64, device="cuda")
    lens = (rp[1:] - rp[:-1]).long()
    rp64 = rp.long()
    step = 4_000_000  # rows per chunk
    for r0 in range(0, m, step):
        r1 = min(m, r0 + step)
        s, e = int(rp64[r0].item()), int(rp64[r1].item())
        if e == s:
            continue
        prod = v[s:e] * x[ci[s:e].long()]
        ax[r0:r1] = torch.segment_reduce(prod, "sum", lengths=lens[r0:r1], unsafe=True)
        mag[r0:r1] = torch.segment_reduce(prod.abs(), "sum", lengths=lens[r0:r1], unsafe=True)
        del prod
    return ax, mag


def _host_checked_reference(torch, A, x, strat):
    """The device reference, guarded against the checker's own faults: torch.segment_reduce has been caught returning wrong
    sums (tools/big_fuzz.py: 25 M mostly empty segments, 1073 rows off by O(1) while every kernel family agreed with a host
    evaluation to the last bits).  Rows where the library (alpha = 1, beta = 0) and the device reference disagree are summed
    once more on the HOST, product by product; the host value replaces the device reference for such a row, so the truth is
    always torch's or the host's arithmetic, never the library's.  (A row where the library is wrong stays wrong against the
    host value and fails the checks that follow.)"""
    m, n, nnz, rp, ci, v = A
    ax, mag = _device_reference(torch, A, x)
    y = _spmv(torch, A, strat, 1.0, 0.0, x, torch.zeros(m, dtype=torch.float64, device="cuda"))
    suspicious = torch.nonzero(((y - ax).abs() / mag.clamp_min(1e-300)) > SCALED_TOL).flatten()
    assert suspicious.numel() <= 5000, ("library and device reference disagree on many rows", int(suspicious.numel()))
    for r in suspicious.tolist():
        a, b = int(rp[r].item()), int(rp[r + 1].item())
        prod = v[a:b].cpu().numpy() * x[ci[a:b].long()].cpu().numpy()
        ax[r] = float(prod.sum())
        mag[r] = float(np.abs(prod).sum())
    if suspicious.numel():
        _say(f"  {int(suspicious.numel())} rows of the device reference re-evaluated on the host")
    return ax, mag


def _verify_y_failures(torch, got, ref):
    """rows breaking the reference benchmark's rule (cli/verification.cpp:15-38), evaluated on the device"""
    d = (got - ref).abs()
    bad = torch.where(ref.abs() <= 1e-12, d >= 1e-14, d / ref.abs() >= 1e-7)
    return int(bad.sum().item())


def _prefix_vs_oracle(torch, oracle, A, strats, x, y0, max_rows=150_000, max_nnz=30_000_000):
    m, n, nnz, rp, ci, v = A
    k = min(max_rows, m)
    if int(rp[k].item()) > max_nnz:  # power-law prefixes: cut by non-zeros instead
        k = max(int(torch.searchsorted(rp, torch.tensor([max_nnz], dtype=rp.dtype, device="cuda")).item()) - 1, 1)
    hrp = rp[: k + 1].cpu().numpy()
    e = int(hrp[-1])
    hci, hv = ci[:e].cpu().numpy(), v[:e].cpu().numpy()
    hx, hy0 = x.cpu().numpy(), y0[:k].cpu().numpy()
    ref = oracle.host_spmv(1.0, 1.0, hrp, hci, hv, hx, hy0)
    for strat in strats:
        y = _spmv(torch, A, strat, 1.0, 1.0, x, y0)[:k].cpu().numpy()
        err = oracle.scaled_error(y, ref, 1.0, 1.0, hrp, hci, hv, hx, hy0)
        assert err <= SCALED_TOL, (strat, "prefix vs oracle", k, err)
        assert oracle.verify_y(y, ref)[2] == 0, (strat, "reference verdict on the prefix")
    return k


def _full_size_checks(torch, oracle, A, strats, seed):
    m, n, nnz, rp, ci, v = A
    x, y0 = _vectors(torch, m, n, seed)
    _say(f"{m} rows, {nnz} nnz: device reference")
    # 1. the reference's protocol (alpha = beta = 1) against the independent device evaluation, all rows
    ax, mag = _host_checked_reference(torch, A, x, strats[0])
    ref = ax + y0
    scale = (mag + y0.abs()).clamp_min(1e-300)
    got = {}
    for strat in strats:
        _say(f"  {strat}: alpha = beta = 1 against the device reference")
        y = _spmv(torch, A, strat, 1.0, 1.0, x, y0)
        err = ((y - ref).abs() / scale).max().item()
        assert err <= SCALED_TOL, (strat, "scaled error vs device reference", err)
        assert _verify_y_failures(torch, y, ref) == 0, (strat, "verify_y rule")
        got[strat] = y
    _say("  alpha / beta decomposition, beta = 0")
    # 2. general alpha / beta decomposes as alpha * (A x) + beta * y0; beta = 0 ignores y
    zeros = torch.zeros(m, dtype=torch.float64, device="cuda")
    for strat in strats:
        full = _spmv(torch, A, strat, 0.5, -2.0, x, y0)
        want = 0.5 * ax - 2.0 * y0
        sc = (0.5 * mag + 2.0 * y0.abs()).clamp_min(1e-300)
        assert ((full - want).abs() / sc).max().item() <= SCALED_TOL, (strat, "alpha/beta")
        nan_y = torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
        y = _spmv(torch, A, strat, 1.0, 0.0, x, nan_y)
        assert ((y - ax).abs() / mag.clamp_min(1e-300)).max().item() <= SCALED_TOL, (strat, "beta = 0 must not read y")
    _say("  row sums, linearity")
    # 3. row sums (x = 1) and linearity in x
    ones = torch.ones(n, dtype=torch.float64, device="cuda")
    sums, sums_mag = _host_checked_reference(torch, A, ones, strats[0])
    x2, _ = _vectors(torch, 1, n, seed + 1)
    for strat in strats:
        y = _spmv(torch, A, strat, 1.0, 0.0, ones, zeros)
        assert ((y - sums).abs() / sums_mag.clamp_min(1e-300)).max().item() <= SCALED_TOL, (strat, "row sums")
        a1 = _spmv(torch, A, strat, 1.0, 0.0, x, zeros)
        a2 = _spmv(torch, A, strat, 1.0, 0.0, x2, zeros)
        both = _spmv(torch, A, strat, 1.0, 0.0, 3.0 * x + x2, zeros)
        sc = (3.0 * mag + _host_checked_reference(torch, A, x2, strats[0])[1]).clamp_min(1e-300)
        assert ((both - (3.0 * a1 + a2)).abs() / sc).max().item() <= 1e-11, (strat, "linearity")
    # 4. a row prefix against the CPU oracle
    _say("  row prefix against the CPU oracle")
    k = _prefix_vs_oracle(torch, oracle, A, strats, x, y0)
    _say(f"  done ({k} prefix rows)")
    return got


@pytest.mark.parametrize("name", synth.SWEEP_NAMES)
def test_configs2_sweep_standin_full_size(torch_dev, oracle, name):
    """BASELINE configs[2]: every stand-in of the large-set sweep at full size, flat (named by BASELINE) and adaptive."""
    torch = torch_dev
    A = synth.sweep_standin_torch(name, device="cuda")
    want = synth.LARGE_SET.get(name) or synth.LARGE_SET_EXTRA[name]
    assert A[:3] == want
    try:
        got = _full_size_checks(torch, oracle, A, ("flat", "adaptive"), seed=0x5EED + synth.SWEEP_NAMES.index(name))
        info = spmv_acc_amd.query_plan(A[3], A[0])
        assert info is not None and info["nnz"] == A[2] and info["flat_tiles"] > 0
        # the two strategies sum in different orders but agree to rounding
        assert (got["flat"] - got["adaptive"]).abs().max().item() <= 1e-10
    finally:
        spmv_acc_amd.release_plans(A[3])
        torch.cuda.empty_cache()


def test_configs3_rmat25_line_enhance_full_size(torch_dev, oracle):
    """BASELINE configs[3]: R-MAT scale 25 (33.5 M rows, ~0.53 B non-zeros, hub rows of millions) under line_enhance --
    the balance probe must hand it to the row-block-plus kernel, which the plan-time timing then replaces by the column-slab
    passes over run lists (k_segment.hip) -- plus flat as the second opinion, and the row-block-plus kernel itself with the passes off."""
    torch = torch_dev
    _say("R-MAT scale 25: generating")
    A = synth.rmat_torch(25, device="cuda", seed=0xC4)
    m, n, nnz, rp = A[0], A[1], A[2], A[3]
    _say(f"R-MAT scale 25: {nnz} nnz")
    assert m == n == 1 << 25 and 480_000_000 < nnz < 537_000_000
    lens = rp[1:] - rp[:-1]
    assert int(lens.max().item()) > 100_000 and int(lens.min().item()) == 0  # power law: hub rows and empty rows
    del lens
    try:
        _full_size_checks(torch, oracle, A, ("line_enhance", "flat"), seed=0xC4C4)
        info = spmv_acc_amd.query_plan(rp, m)
        assert info["plus_blocks"] > 0  # line_enhance was rescued by the row-block preprocessing pass ...
        # ... and (round 3) the plan-time timing then preferred the column-slab passes over run lists: that is the path checked above
        assert info["slab_passes"] >= 2, info
        # the one-kernel path it was timed against (hinted row-block-plus), same checks
        spmv_acc_amd.release_plans(rp)
        spmv_acc_amd.load_library().spmv_acc_set_tunable(b"slab_segments", 0)
        _full_size_checks(torch, oracle, A, ("line_enhance",), seed=0xC4C4)
        assert spmv_acc_amd.query_plan(rp, m)["slab_passes"] == 0
    finally:
        spmv_acc_amd.load_library().spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans(rp)
        del A
        torch.cuda.empty_cache()


def test_configs3_rmat25_out_of_place_and_column_slabs_full_size(torch_dev, oracle, hiplib):
    """Round 3 at configs[3]'s full size: the out-of-place entry is bitwise the in-place one (line_enhance, i.e. the row-block-plus
    kernel with long rows sliced over blocks and folded by its second kernel); the opt-in column slabs (col_slabs = 8: eight
    consecutive SpMVs over the plan's re-ordered copy, the first applying beta) agree with the independent device evaluation on
    every row and with the CPU oracle on a row prefix; values edited in place are picked up by spmv_acc_refresh_values."""
    torch = torch_dev
    A = synth.rmat_torch(25, device="cuda", seed=0xC4)
    m, n, nnz, rp, ci, v = A
    x, y0 = _vectors(torch, m, n, 0xC4C5)
    try:
        # (settled first: the calls before that are served by the plan's rule twin -- round 6 -- and would differ from the settled plan's in the last bits)
        hiplib.spmv_acc_set_tunable(b"col_slabs", 0)  # (the passes / the one-kernel path: prepare would otherwise build the automatic slab-major copy)
        spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy="line_enhance", beta=-2.0)
        inplace = _spmv(torch, A, "line_enhance", 0.5, -2.0, x, y0)
        y_in, y_out = y0.clone(), torch.full((m,), float("nan"), dtype=torch.float64, device="cuda")
        spmv_acc_amd.csr_spmv(0.5, -2.0, m, n, nnz, rp, ci, v, x, y_out, strategy="line_enhance", y_in=y_in)
        torch.cuda.synchronize()
        assert torch.equal(y_out, inplace) and torch.equal(y_in, y0)
        del y_in, y_out, inplace
        ax, mag = _host_checked_reference(torch, A, x, "line_enhance")
        spmv_acc_amd.release_plans(rp)
        hiplib.spmv_acc_set_tunable(b"col_slabs", 8)
        _say("R-MAT 25 with 8 column slabs")
        y = _spmv(torch, A, "line_enhance", 1.0, 1.0, x, y0)
        ref = ax + y0
        scale = (mag + y0.abs()).clamp_min(1e-300)
        assert ((y - ref).abs() / scale).max().item() <= SCALED_TOL, "column slabs vs device reference"
        assert _verify_y_failures(torch, y, ref) == 0
        y = _spmv(torch, A, "line_enhance", 0.5, 0.0, x, torch.full((m,), float("nan"), dtype=torch.float64, device="cuda"))
        assert ((y - 0.5 * ax).abs() / (0.5 * mag).clamp_min(1e-300)).max().item() <= SCALED_TOL, "column slabs, beta = 0"
        _prefix_vs_oracle(torch, oracle, A, ("line_enhance",), x, y0)
        # values edited in place (scaled by -2): one refresh pass, same plans
        plans = hiplib.spmv_acc_cached_plans()
        v.mul_(-2.0)
        assert spmv_acc_amd.refresh_values(rp) == 1
        y = _spmv(torch, A, "line_enhance", 1.0, 1.0, x, y0)
        assert ((y - (y0 - 2.0 * ax)).abs() / (2.0 * mag + y0.abs()).clamp_min(1e-300)).max().item() <= SCALED_TOL, "after refresh_values"
        assert hiplib.spmv_acc_cached_plans() == plans
    finally:
        hiplib.spmv_acc_reset_tunables()
        spmv_acc_amd.release_plans(rp)
        del A
        torch.cuda.empty_cache()


def test_configs4_banded_shard_full_size(torch_dev, oracle):
    """BASELINE configs[4]: rank 3's shard (32 M rows, 256 M non-zeros, global column ids into a 256 M-entry x) of the
    256 M-row banded matrix.  Closed form for x = 1, the generic full-size checks, and the first rows against the oracle."""
    torch = torch_dev
    rows, total, first = 32_000_000, 256_000_000, 96_000_000
    rp, ci, v = synth.banded_torch(rows, first_row=first, total_rows=total, device="cuda")
    nnz = int(rp[-1].item())
    assert nnz == 8 * rows
    A = (rows, total, nnz, rp, ci, v)
    offs = np.arange(-4, 4)
    want2 = [float(np.sum(np.where((r + offs) % 2 == 0, 1.0, -1.0) / (1.0 + np.abs(offs)))) for r in (first, first + 1)]
    strats = ("adaptive", "flat", "line_enhance")
    try:
        ones = torch.ones(total, dtype=torch.float64, device="cuda")
        zeros = torch.zeros(rows, dtype=torch.float64, device="cuda")
        for strat in strats:
            y = _spmv(torch, A, strat, 1.0, 0.0, ones, zeros)
            assert (y[0::2] - want2[0]).abs().max().item() <= 1e-14 and (y[1::2] - want2[1]).abs().max().item() <= 1e-14, strat
        del ones, zeros
        _full_size_checks(torch, oracle, A, strats, seed=0xC5)
    finally:
        spmv_acc_amd.release_plans(rp)
        torch.cuda.empty_cache()


def test_bench_line_carries_every_baseline_config(torch_dev):
    """The driver's command -- `python bench.py` with its defaults -- prints ONE SHORT JSON line on stdout (< 6000 bytes: the driver keeps
    an 8 KB tail; one short row per measurement, as statistics_logger.cpp:11-31 prints) whose headline is configs[1] with `roofline` and
    `cpu_baseline`, plus one figure per extra leg: configs[2] (`sweep`: 12 stand-ins -> [flat, adaptive] fractions), configs[3]
    (`rmat25`), configs[4] (`banded_shard`).  Every detail (both protocols per leg, the tile kernel alone, opt-in legs, notes) is in
    bench_full.json beside the script."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full_path = os.path.join(root, "bench_full.json")
    if os.path.exists(full_path):
        os.remove(full_path)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "50", "--warmup", "5", "--cpu-seconds", "2"],
                         capture_output=True, text=True, check=True).stdout.strip().splitlines()
    assert len(out) == 1, out[:3]  # libraries' chatter goes to stderr
    assert len(out[0]) < 6000, len(out[0])
    line = json.loads(out[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["config"]["strategy"] == "adaptive" and "Hardesty3" in line["config"]["workload"]
    assert line["dtype"] == "f64" and line["value"] > 0 and line["ms_per_step"] > 0
    rl = line["roofline"]
    assert rl["bound"] == "hbm" and rl["peak"] == 8000.0 and abs(rl["frac"] - rl["achieved"] / rl["peak"]) < 1e-3
    assert rl["traffic"] is not None and rl["traffic_lower_bound"] is not None and rl["back_to_back"]["frac"] > 0
    assert abs(rl["achieved"] - rl["algorithmic_bytes_per_launch"] / (rl["launch_ms_mean"] * 1e-3) / 1e9) < 1.0
    assert line["cpu_baseline"]["kind"] == "port" and line["cpu_baseline"]["cores"] >= 1 and line["cpu_baseline"]["value"] > 0
    assert set(line["sweep"]) == set(synth.SWEEP_NAMES)
    assert all(len(v) == 2 and 0.05 < v[0] < 1.2 and 0.05 < v[1] < 1.2 for v in line["sweep"].values())
    for strat in ("flat", "adaptive"):  # the gate's count in both regimes, side by side
        assert 0 <= line["sweep_summary"][strat]["ge_0.70"] <= 12 and 0 <= line["sweep_summary"][strat]["ge_0.70_in_process"] <= 12
    assert line["rmat25"]["nnz"] > 480_000_000 and line["rmat25"]["us"] > 1000 and line["rmat25"]["frac"] > 0
    assert 0.3 < line["banded_shard"]["frac"] < 1.2
    assert line["first_call_ms"] > 0 and line["details"] == "bench_full.json"

    d = json.load(open(full_path))  # the full record
    assert d["value"] == line["value"] and d["roofline"]["frac"] == line["roofline"]["frac"]
    def both_protocols(leg, tag):
        # every figure is quoted on the reference harness's per-launch protocol (y reset, median) with the back-to-back mean beside it
        assert leg["per_launch_reset_ms_median"] > 0 and leg["back_to_back_ms_mean"] > 0, (tag, leg)
        assert abs(leg["us"] - 1e3 * leg["per_launch_reset_ms_median"]) <= 0.011, (tag, leg)  # us / frac follow the former
        assert leg["frac_back_to_back"] > 0

    for name, row in d["sweep"].items():
        for strat in ("flat", "adaptive"):
            assert row[strat]["us"] > 0 and 0.05 < row[strat]["frac"] < 1.2, (name, strat, row[strat])
            both_protocols(row[strat], (name, strat))
            assert line["sweep"][name][("flat", "adaptive").index(strat)] == row[strat]["frac"]
    assert set(d["sweep_in_process"]) == set(synth.SWEEP_NAMES)
    assert "ge_0.70" in d["sweep_summary"]["flat"] and "ge_0.70_back_to_back" in d["sweep_summary"]["flat"]
    # `flat` as shipped may run the row-block kernel on balanced rows (timed per matrix): the tile kernel alone is reported beside it
    assert all(row["flat_tile_kernel"]["us"] > 0 for row in d["sweep"].values()) and "ge_0.70" in d["sweep_summary"]["flat_tile_kernel"]
    assert "child process" in d["legs_measured"]
    both_protocols(d["rmat25"]["line_enhance"], "rmat25")
    # (restored in round 5: the default path -- the slab passes the plan-time timing chose -- beats the one-kernel path it was timed against)
    both_protocols(d["rmat25"]["line_enhance_without_slab_passes"], "rmat25, one-kernel path")
    both_protocols(d["rmat25"]["line_enhance_slab_passes_only"], "rmat25, run-list passes only")
    assert "path" in d["rmat25"] and ("slab-major copy" in d["rmat25"]["path"] or "column-slab passes" in d["rmat25"]["path"])
    # (round 6: the default path is the faster of the copy and the passes -- never slower than the passes alone beyond noise -- and both beat the one-kernel path)
    assert d["rmat25"]["line_enhance"]["us"] <= 1.02 * d["rmat25"]["line_enhance_slab_passes_only"]["us"], d["rmat25"]
    assert d["rmat25"]["line_enhance_slab_passes_only"]["us"] < d["rmat25"]["line_enhance_without_slab_passes"]["us"], d["rmat25"]
    assert d["plan"]["settled"] is True  # every timed figure is taken on a settled plan (spmv_acc_query_plan_settled)
    assert d["banded_shard"]["rows"] == 32_000_000
    both_protocols(d["banded_shard"]["adaptive"], "banded_shard")
    assert d["per_launch_reset_ms_median"] > 0 and d["back_to_back_ms_mean"] > 0
    # roofline.frac: the kernel's average launch duration over the timed region (round 5: what the rocprofv3 summary of the command averages to);
    # the reference harness's event pair -- kernel + the protocol's floor -- and the kernel clock under that protocol beside it
    assert abs(d["roofline"]["launch_ms_mean"] - d["ms_per_step_events"]) < 1e-9 and d["roofline"]["launch_ms_mean"] <= d["ms_per_step"]
    assert abs(d["roofline"]["kernel_clock_reset_protocol"]["launch_ms_median"] - d["kernel_clock_ms_median"]) < 1e-9
    assert abs(d["roofline"]["per_launch_protocol"]["launch_ms_median"] - d["per_launch_reset_ms_median"]) < 1e-9
    # (the kernel clock and the event pair are medians of two separate series of launches: the kernel's own time is below the pair's up to their noise)
    assert d["kernel_clock_ms_median"] <= 1.02 * d["per_launch_reset_ms_median"] and d["roofline"]["per_launch_protocol"]["frac"] <= 1.01 * d["roofline"]["frac"]
    assert line["roofline"]["per_launch_protocol"]["frac"] == d["roofline"]["per_launch_protocol"]["frac"]
    assert d["region_reps"] >= 5 and len(d["ms_per_step_wall_all"]) == d["region_reps"]
    for name, row in d["sweep"].items():
        for strat in ("flat", "adaptive"):
            assert 0 < row[strat]["us_kernel_clock"] <= 1.05 * row[strat]["us"] + 0.3 and row[strat]["launches_per_spmv"] >= 1, (name, strat, row[strat])
    assert "ge_0.70_kernel_clock" in d["sweep_summary"]["flat"] and "ge_0.70_kernel_clock" in line["sweep_summary"]["adaptive"]
    cb = d["cpu_baseline"]
    assert cb["bitwise_equal_to_sequential"] and cb["stream_triad_gbs"] > 0 and len(cb["value_median_per_round"]) == 3 and cb["cores"] >= 1
    assert "builder-run" in d["roofline"]["traffic_source"]
    # round 6: the frozen definition travels with the figure, and a cold-cache column stands beside every fraction (context, never a gate)
    assert line["roofline"]["definition"] == {"version": "r05", "frac": "back_to_back", "legs_and_gates": "per_launch"}
    assert 0 < d["roofline"]["frac_cold"] <= 1.02 * d["roofline"]["per_launch_protocol"]["frac"], d["roofline"]
    assert line["roofline"]["frac_cold"] == d["roofline"]["frac_cold"] and -0.05 < d["roofline"]["cached_share_of_frac"] < 0.6
    for name, row in d["sweep"].items():
        for strat in ("flat", "adaptive"):
            assert 0 < row[strat]["frac_cold"] <= 1.03 * row[strat]["frac"] + 0.01, (name, strat, row[strat])  # a cold start is never faster (up to noise)
            assert row[strat]["col16"] in (0, 16, 32, 64), (name, strat, row[strat])
        assert row["adaptive_colindex_only"]["col16"] == 0 and row["adaptive_colindex_only"]["us"] > 0
    for strat in ("flat", "adaptive"):
        assert 0 <= line["sweep_summary"][strat]["ge_0.70_cold"] <= line["sweep_summary"][strat]["ge_0.70_kernel_clock"] + 1
        assert 0 <= line["sweep_summary"][strat]["stand_ins_on_16_bit_columns"] <= 12
    assert d["rmat25"]["line_enhance"]["frac_cold"] > 0 and line["banded_shard"]["frac_cold"] > 0
