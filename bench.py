#!/usr/bin/env python3
"""bench.py -- fp64 CSR SpMV (y = alpha*A*x + beta*y, alpha = beta = 1) on MI355X through the C ABI.

    python bench.py --gpus N --steps K --warmup W

A "step" is one SpMV over one synthetic matrix resident in HBM.  N = 1 runs BASELINE.json configs[1]
(Hardesty3-sized stand-in, 8,217,820 x 7,591,564, 40,451,632 nnz, adaptive strategy).  N > 1: one rank per GPU
(torch.distributed over RCCL): every rank owns one such matrix as its row range of an (N*m) x n global matrix
(weak scaling), x is replicated, and each step ends with the RCCL allgather of the y sub-vectors.  Launched either by
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...` or plainly as `python bench.py --gpus N ...`:
without WORLD_SIZE in the environment the process starts the N ranks itself as CHILD processes (before anything touches
the GPU -- the parent imports neither torch nor the library), relays rank 0's line and exits with the children's status.
Rank 0 prints ONE JSON line.

value       = 2 * nnz_total * K / wall_seconds / 1e9  [GFLOP/s], wall-clock over exactly K back-to-back steps bracketed by
              barrier + torch.cuda.synchronize() on both sides, max over ranks; the region is repeated REGION_REPS times inside the one
              command and wall_seconds is the MEDIAN repetition (the reference reports medians, benchmark_time.cpp:23-43); the
              event time of the same regions stands beside it (ms_per_step_events).
roofline    = algorithmic bytes (12*nnz + 4*(m+1) + 8*n + 16*m) / the dominant kernel's average launch duration over the
              TIMED REGION (hipEvents on the library stream around the K launches, median repetition): the figure the committed rocprofv3
              --kernel-trace --stats summary of the same command averages to.  Beside it: `per_launch_protocol` -- the REFERENCE HARNESS'S
              protocol (benchmark/csr_spmv.hpp:66-74: y reset by a device copy before every launch, one event pair per launch, median;
              it also holds the protocol's floor of marker packets + dispatch latency), the figure every sweep gate is counted on -- and
              `kernel_clock_reset_protocol`: the kernel's own start / stop events (spmv_acc_time_spmv_kernels) under that protocol.
cpu_baseline= the oracle (CPU restatement of cli/verification.cpp:56-66) on the host cores, same matrix.

N = 1, default: the extra legs (configs[2] sweep, configs[3] R-MAT 25, configs[4] banded shard) are measured in one CHILD process per matrix, as the
reference's batch runs its CLI once per matrix file (examples/large-data-set-batch.sh); `--legs-in-process` keeps them here.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling

# ---- the clocks, FROZEN as of round 5 (tests/test_bench_launch.py::test_roofline_frac_keeps_its_source pins this table and
# roofline_block below: a figure may move only because a kernel's microseconds moved, never because a definition did) -----------------------------
#   back_to_back      one hipEvent pair around the K launches of the timed region, y never reset, median of REGION_REPS regions / K
#                     = `value`, `ms_per_step`, `roofline.achieved`, `roofline.frac` (since round 5; rounds 1-4 quoted per_launch on that key:
#                     their lines' roofline.frac is this record's roofline.per_launch_protocol.frac)
#   per_launch        the reference harness's protocol (benchmark/csr_spmv.hpp:66-74, benchmark/utils/benchmark_time.cpp:23-43): y restored by a
#                     non-temporal device copy before EVERY launch, outside the event pair; one default hipEvent pair per launch; median
#                     = every extra leg's `us` / `frac` and every `ge_0.70` count
#   kernel_clock      the same launches' own start / stop timestamps (hipExtLaunchKernelGGL), summed per call, median
#   cold              per_launch with 1 GiB of default-policy scratch traffic between the y reset and the start event (round 6): the launch starts with
#                     nothing of the matrix in the L2s or the 256 MB Infinity Cache.  Context (`*_cold` keys), never a gate.
ROOFLINE_DEFINITION = {
    "version": "r05",
    "frac": "back_to_back",
    "achieved": "algorithmic bytes (12 nnz + 4 (m + 1) + 8 n + 16 m; 8 m for y at beta = 0) / back_to_back launch time",
    "legs_and_gates": "per_launch",
    "earlier_rounds": "rounds 1-4 quoted roofline.frac on per_launch: compare their lines with roofline.per_launch_protocol.frac (r04's 0.5461 -> this key)",
}
FLUSH_BYTES = 1 << 30


def roofline_block(b_alg, b2b_ms, kernel_ms, ev_ms, cold=None):
    """The roofline object of the line from the four clocks (milliseconds per launch).  `frac` / `achieved` come from b2b_ms and from nothing else."""
    gbs = lambda ms: round(b_alg / (ms * 1e-3) / 1e9, 2)  # noqa: E731
    fr = lambda ms: round(b_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)  # noqa: E731
    out = {"bound": "hbm", "achieved": gbs(b2b_ms), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs(b2b_ms) / HBM_PEAK_GBS, 4),
           "definition": dict(ROOFLINE_DEFINITION),
           "kernel_clock_reset_protocol": {"what": "the kernel's own start / stop events (hipExtLaunchKernelGGL), y reset by a device copy before each launch, median",
                                           "launch_ms_median": round(kernel_ms, 6), "achieved": gbs(kernel_ms), "frac": fr(kernel_ms)},
           "per_launch_protocol": {"what": "event pair around each call, y reset before it, median (benchmark/csr_spmv.hpp:66-74): kernel + the protocol's floor",
                                   "launch_ms_median": round(ev_ms, 6), "achieved": gbs(ev_ms), "frac": fr(ev_ms)},
           "algorithmic_bytes_per_launch": b_alg, "launch_ms_mean": round(b2b_ms, 6),
           "back_to_back": {"launch_ms_mean": round(b2b_ms, 6), "achieved": gbs(b2b_ms), "frac": fr(b2b_ms)}}
    if cold is not None:
        out["cold_protocol"] = {"what": f"per-launch protocol with {FLUSH_BYTES >> 20} MiB of default-policy scratch traffic before every timed launch: nothing of the "
                                        "matrix in the L2s or the Infinity Cache when it starts (context, never a gate)",
                                "launch_ms_median": round(cold, 6), "achieved": gbs(cold), "frac": fr(cold)}
        out["frac_cold"] = fr(cold)
        out["cached_share_of_frac"] = round(1.0 - fr(cold) / fr(ev_ms), 4)  # how much of the per-launch figure the caches' carry-over is worth
    return out


_FLUSH = {}


def flush_buffer(torch, device):
    """1 GiB of device scratch for the cold protocol (once per process; None if it cannot be had)."""
    key = str(device)
    if key not in _FLUSH:
        try:
            _FLUSH[key] = torch.zeros(FLUSH_BYTES, dtype=torch.uint8, device=device)
        except RuntimeError:
            _FLUSH[key] = None
    return _FLUSH[key]


def cold_ms(torch, strat, A, x, y, y0, iters, alpha=1.0, beta=1.0):
    """Median per-launch time under the cold protocol (spmv_acc_time_spmv_cold), or None without the scratch buffer."""
    import spmv_acc_amd

    m, n, nnz, rp, ci, v = A
    flush = flush_buffer(torch, x.device)
    if flush is None:
        return None
    per = spmv_acc_amd.time_spmv_cold(strat, iters, alpha, beta, m, n, nnz, rp, ci, v, x, y, y0, flush)
    return float(np.median(per))
REGION_REPS = 7        # `value` / `ms_per_step`: median over this many repetitions of the K-step region (benchmark_time.cpp:23-43: the reference reports a median)
T_START = time.perf_counter()


def median_region(run, reps, sync_all, reduce_max=None):
    """`reps` repetitions of ONE timed region -- `run()` = exactly K steps --, each bracketed by barrier + device synchronisation on both sides
    (`sync_all`).  Returns (median wall seconds, the per-repetition wall seconds).  N > 1: `reduce_max` takes the MAX over ranks of every
    repetition first, so that all ranks agree on the list and on its median."""
    walls = []
    for _ in range(reps):
        sync_all()
        t0 = time.perf_counter()
        run()
        sync_all()
        walls.append(time.perf_counter() - t0)
    if reduce_max is not None:
        walls = reduce_max(walls)
    return float(np.median(walls)), [float(w) for w in walls]


def budget_left(budget_s):
    """Seconds left of this process's wall-clock allowance (N > 1: SPMV_ACC_BENCH_BUDGET_S, default 420 -- the side legs of a multi-GPU run
    are skipped rather than allowed to eat the driver's timeout when an exchange turns out slow)."""
    return budget_s - (time.perf_counter() - T_START)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--workload", default="hardesty3",
                   help="hardesty3 | banded | rmat | one of the large-set names (boneS10, Bump_2911, ...)")
    p.add_argument("--strategy", default=None, help="KERNEL_STRATEGY name (default: adaptive; banded/rmat per BASELINE)")
    p.add_argument("--scale", type=float, default=1.0, help="shrink the workload (for rehearsals only)")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-sensitivity", action="store_true", help="skip the far_fraction=0 variant of the N=1 workload")
    p.add_argument("--no-legs", action="store_true",
                   help="N = 1: skip the extra live legs (configs[2] sweep, configs[3] R-MAT 25, configs[4] banded shard)")
    p.add_argument("--cpu-seconds", type=float, default=12.0)
    p.add_argument("--exchange", default="auto", choices=["auto", "allgather", "p2p", "ghost"],
                   help="N > 1: how the y slices travel -- RCCL all_gather_into_tensor, a direct point-to-point fan-out "
                        "over all xGMI links, or (auto) whichever is faster on this job's communicator, timed before the run; "
                        "ghost (square workloads, i.e. --workload banded): x is partitioned like the rows and each rank receives "
                        "only the entries its columns reference (x <- alpha*A*x iteration, beta = 0)")
    p.add_argument("--no-overlap", action="store_true", help="N>1: wait for each allgather before the next SpMV")
    p.add_argument("--legs-in-process", action="store_true",
                   help="N = 1: measure the extra legs inside this process instead of one child process per matrix (e.g. to see their kernels "
                        "in one rocprofv3 trace)")
    p.add_argument("--leg-child", default=None, help=argparse.SUPPRESS)  # internal: measure ONE leg and print its JSON (see extra_legs)
    p.add_argument("--cpu-baseline-child", default=None, help=argparse.SUPPRESS)  # internal: the timed CPU runs over the arrays in this directory
    return p.parse_args()


def build_workload(args, torch, device, rank):
    from spmv_acc_amd import synth

    w = args.workload
    if w == "hardesty3":
        m, n, nnz, rp, ci, v = synth.hardesty3_like_torch(device=device, seed=0xC2 + rank, scale=args.scale)
        return dict(name="Hardesty3-like (SuiteSparse Hardesty3 dims, synthetic stand-in)", m=m, n=n, nnz=nnz, rp=rp,
                    ci=ci, v=v, strategy=args.strategy or "adaptive")
    if w == "banded":
        rows = int(32_000_000 * args.scale)
        world = max(args.gpus, 1)
        rp, ci, v = synth.banded_torch(rows, first_row=rank * rows, total_rows=world * rows, device=device)
        return dict(name="banded offsets -4..+3 (BASELINE configs[4] shard)", m=rows, n=world * rows,
                    nnz=int(rp[-1].item()), rp=rp, ci=ci, v=v, strategy=args.strategy or "adaptive", global_cols=True)
    if w == "rmat":
        scale = max(int(round(25 + np.log2(max(args.scale, 1e-9)))), 8)
        m, n, nnz, rp, ci, v = synth.rmat_torch(scale, device=device, seed=0xC4 + rank)
        return dict(name=f"R-MAT scale {scale} edge factor 16", m=m, n=n, nnz=nnz, rp=rp, ci=ci, v=v,
                    strategy=args.strategy or "line_enhance")
    if w in synth.LARGE_SET or w in synth.LARGE_SET_EXTRA:
        m, n, nnz, rp, ci, v = synth.large_set_like_torch(w, device=device, seed=0xC300 + rank, scale=args.scale)
        return dict(name=f"{w}-like (synthetic stand-in)", m=m, n=n, nnz=nnz, rp=rp, ci=ci, v=v,
                    strategy=args.strategy or "flat")
    raise SystemExit(f"unknown workload {w}")


def host_cpus():
    """CPUs this process may really use: the affinity mask, cut by the cgroup's CPU quota where there is one (a box that shows 128 hardware
    threads but grants 16 CPUs' worth of time throttles 128 busy threads -- one source of the 2x swings of earlier rounds)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = max(1, int(float(q) / float(period) + 0.5))
    except (OSError, ValueError):
        pass
    return (min(n, quota) if quota else n), n, quota


def physical_cores():
    """Physical cores among the CPUs this process may run on (one per set of SMT siblings)."""
    cpus = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    seen = set()
    for c in cpus:
        try:
            seen.add(open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip())
        except OSError:
            seen.add(str(c))
    return max(len(seen), 1)


def cpu_baseline_child(args):
    """`python bench.py --cpu-baseline-child DIR`: the timed CPU runs, in a process of their own so that OMP_PROC_BIND / OMP_PLACES are in the
    environment BEFORE the OpenMP runtime is loaded (this process never imports torch).  Prints one JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    d = args.cpu_baseline_child
    rp, ci, v, hx, hy0 = (np.load(os.path.join(d, f"{k}.npy"), mmap_mode="r") for k in ("rp", "ci", "v", "x", "y0"))
    m, n, nnz = len(rp) - 1, len(hx), int(rp[-1])
    seconds = args.cpu_seconds
    threads, affinity, quota = host_cpus()
    threads = int(os.environ.get("SPMV_ACC_CPU_THREADS", threads))
    b_alg = 12 * nnz + 4 * (m + 1) + 8 * n + 16 * m
    # one thread: the sequential form as the reference's verification path runs it (cli/verification.cpp:56-66)
    t1 = float("inf")
    y_seq = None
    t_begin = time.perf_counter()
    for _ in range(3):
        y = np.array(hy0)
        t0 = time.perf_counter()
        oracle_lib.host_spmv_inplace(1.0, 1.0, rp, ci, v, hx, y)
        t1 = min(t1, time.perf_counter() - t0)
        y_seq = y
        if time.perf_counter() - t_begin > 0.25 * seconds:
            break
    triad = oracle_lib.stream_triad_gbs(1 << 26, threads, 5)  # 3 x 512 MiB
    # all threads: placed arrays (first touch by the reading threads), y restored before every run; three independent rounds
    probe, _ = oracle_lib.host_spmv_bench(1.0, 1.0, rp, ci, v, hx, hy0, threads, 3)
    per_round = max(5, min(200, int(0.2 * seconds / max(float(np.min(probe)), 1e-6))))
    rounds, same = [], True
    for _ in range(3):
        secs, y_par = oracle_lib.host_spmv_bench(1.0, 1.0, rp, ci, v, hx, hy0, threads, per_round)
        rounds.append(secs)
        same = same and bool(np.array_equal(y_par, y_seq))  # every row is summed left to right in both forms: bitwise equal
    best = float(min(np.min(r) for r in rounds))
    meds = [float(np.median(r)) for r in rounds]
    typical = float(np.median(meds))  # `value`: the median round's median run (best-of-N beside it: on a shared host the best run is an outlier)
    cpu_model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    gf = lambda t: round(2.0 * nnz / t / 1e9, 3)  # noqa: E731
    print(json.dumps({
        "value": gf(typical), "value_best": gf(best), "unit": "GFLOP/s", "cores": threads, "kind": "port",
        "sample": f"whole matrix ({m} rows, {nnz} nnz), median run of 3 x {per_round} on {threads} pinned threads (OMP_PROC_BIND={os.environ.get('OMP_PROC_BIND')}, "
                  f"OMP_PLACES={os.environ.get('OMP_PLACES')}), arrays first-touched by the threads that read them, y restored before every run; 1 thread: best of 3",
        "cpu_model": cpu_model, "value_1thread": gf(t1),
        "value_median_per_round": [gf(t) for t in meds], "spread_of_round_medians": round(max(meds) / min(meds) - 1.0, 4),
        "achieved_gbs": round(b_alg / typical / 1e9, 1), "stream_triad_gbs": round(triad, 1),
        "frac_of_stream_triad": round(b_alg / typical / 1e9 / triad, 4) if triad > 0 else None,
        "hardware_threads": affinity, "cgroup_cpu_quota": quota, "bitwise_equal_to_sequential": same}))


def cpu_baseline(W, x, y0, seconds):
    """Oracle timed on the host cores (rank 0, N = 1 only): whole matrix, ~`seconds` of CPU work, in a CHILD process that never loads torch
    and has the OpenMP placement variables in its environment from the start (cpu_baseline_child).  The arrays travel through /dev/shm."""
    import shutil
    import subprocess
    import tempfile

    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    d = tempfile.mkdtemp(prefix="spmv_acc_cpu_baseline_", dir=base)
    try:
        for k, t in (("rp", W["rp"]), ("ci", W["ci"]), ("v", W["v"]), ("x", x), ("y0", y0)):
            np.save(os.path.join(d, f"{k}.npy"), t.cpu().numpy())
        # Two placements, a process each (libgomp reads the placement variables when it is loaded): (A) one thread per PHYSICAL core of the host,
        # spread over both sockets -- what the host can do; on the GPU boxes of this pool that is 128 threads under a cgroup CPU quota of 16, i.e.
        # short bursts that the quota allows but does not sustain -- and (B) as many threads as the quota grants, packed (close).  `value` is the
        # faster of the two; both are in the record (profiles/r05_cpu_placement_probe.txt: A 34-36 GFLOP/s +-3 %, B 11.4-11.8 +-2 %; unpinned 128
        # threads over numpy-allocated arrays -- rounds 1-4 -- read 20-40 from run to run).
        quota_threads = host_cpus()[0]
        configs = [("all physical cores, spread", physical_cores(), "spread"), ("the cgroup's CPU quota, close", quota_threads, "close")]
        if configs[0][1] <= configs[1][1]:
            configs = configs[1:]
        runs = []
        for label, threads, bind in configs:
            env = dict(os.environ, OMP_PROC_BIND=bind, OMP_PLACES="cores", OMP_NUM_THREADS=str(threads), OMP_DYNAMIC="false", SPMV_ACC_CPU_THREADS=str(threads))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", d, "--cpu-seconds", str(seconds / len(configs))],
                               env=env, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                runs.append(dict(json.loads(lines[-1]), placement=label))
            else:
                print(f"[bench cpu_baseline] {label}: child failed: {(r.stderr or r.stdout)[-300:]!r}", file=sys.stderr)
        if not runs:
            return {"value": None, "unit": "GFLOP/s", "cores": quota_threads, "kind": "port", "sample": "failed"}
        # `value`: the faster placement (on these hosts always the spread one: 35-40 GFLOP/s over five runs; the packed one 8-12.4); choosing by the
        # rounds' agreement instead made `value` jump between the two placements from run to run
        pick = max(runs, key=lambda r: r["value"])
        out = dict(pick)
        out["placements"] = [{k: r[k] for k in ("placement", "cores", "value", "value_best", "value_median_per_round", "spread_of_round_medians",
                                                "stream_triad_gbs", "frac_of_stream_triad")} for r in runs]
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def pmc_traffic(workload, strategy, key="corrected_bytes"):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json, regenerated from the PMC
    text by tools/profile_round.sh), or None.  `corrected_bytes` is the guide's (2*FETCH + WRITE)*1024 -- an upper bound
    where gathers miss L2 --, `lower_bound_bytes` counts every L2-missing gather at its 64-B sector."""
    try:
        table = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        return table[f"{workload}|{strategy}"][key]
    except (OSError, KeyError, ValueError):
        return None


def two_protocols(torch, strat, A, x, y, y0, iters_reset, iters_b2b, alpha=1.0, beta=1.0):
    """The two timings every number in the line is quoted on, in milliseconds per launch:
    per_launch_reset_ms_median -- the reference harness's protocol (benchmark/csr_spmv.hpp:66-74): y restored from y0 by a
        device copy before every launch (outside the event pair), one hipEvent pair per launch on the library stream, median;
    back_to_back_ms_mean -- one event pair around `iters_b2b` launches, y never reset: each launch starts in what the previous
        one left in L2 / the Infinity Cache (and the library walks the matrix in alternating directions to use that)."""
    import spmv_acc_amd

    m, n, nnz, rp, ci, v = A
    per = spmv_acc_amd.time_spmv(strat, iters_reset, alpha, beta, m, n, nnz, rp, ci, v, x, y, y0=y0)
    y.copy_(y0)
    b2b = spmv_acc_amd.time_spmv_total(strat, iters_b2b, alpha, beta, m, n, nnz, rp, ci, v, x, y) / iters_b2b
    return float(np.median(per)), float(b2b), float(np.min(per))


def nofence_ms(strat, A, x, y, y0, iters, alpha=1.0, beta=1.0):
    """The per-launch protocol with events created hipEventDisableSystemFence (HIP's documented form for events that only
    measure time: no cache write-back / invalidation inside the event).  Shown beside the default-event figure, never instead."""
    import spmv_acc_amd

    m, n, nnz, rp, ci, v = A
    per = spmv_acc_amd.time_spmv(strat, iters, alpha, beta, m, n, nnz, rp, ci, v, x, y, y0=y0,
                                 event_flags=spmv_acc_amd.EVENT_DISABLE_SYSTEM_FENCE)
    return float(np.median(per))


def timed_leg(torch, strat, A, x, y0, iters, warm=10, beta=1.0, cols_touched=None):
    """One extra leg: `warm` untimed launches (the first builds the plan), then both protocols (two_protocols).
    us / frac: the reference's per-launch protocol with y reset (what every gate is quoted on); us_back_to_back /
    frac_back_to_back beside them.  frac = algorithmic bytes / time / 8 TB/s."""
    import spmv_acc_amd
    from spmv_acc_amd import synth

    m, n, nnz, rp, ci, v = A
    y = y0.clone()
    spmv_acc_amd.prepare(m, n, nnz, rp, ci, v, x, strategy=strat, beta=beta)  # every per-matrix timing up front: none is left to fall into the timed launches
    for _ in range(warm):
        spmv_acc_amd.csr_spmv(1.0, beta, m, n, nnz, rp, ci, v, x, y, strategy=strat)
    torch.cuda.synchronize()
    y.copy_(y0)
    reset_ms, b2b_ms, _ = two_protocols(torch, strat, A, x, y, y0, max(20, iters // 3), iters, beta=beta)
    nf_ms = nofence_ms(strat, A, x, y, y0, 20, beta=beta)
    _, kn, kl = spmv_acc_amd.time_spmv_kernels(strat, max(20, iters // 3), 1.0, beta, m, n, nnz, rp, ci, v, x, y, y0=y0)
    kn_ms = float(np.median(kn))
    c_ms = cold_ms(torch, strat, A, x, y, y0, max(8, min(20, iters // 3)), beta=beta)
    # (a row shard with global column ids reads only the columns its rows reference, not all n entries of x)
    b = synth.algorithmic_bytes(m, n if cols_touched is None else cols_touched, nnz, beta_nonzero=beta != 0.0)
    info = spmv_acc_amd.query_plan(rp, m) or {}
    frac = lambda ms: round(b / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)  # noqa: E731
    return {"us": round(reset_ms * 1e3, 2), "frac": frac(reset_ms),
            "per_launch_reset_ms_median": round(reset_ms, 6), "back_to_back_ms_mean": round(b2b_ms, 6),
            "us_back_to_back": round(b2b_ms * 1e3, 2), "frac_back_to_back": frac(b2b_ms),
            "us_events_without_system_fence": round(nf_ms * 1e3, 2), "frac_events_without_system_fence": frac(nf_ms),
            "us_kernel_clock": round(kn_ms * 1e3, 2), "frac_kernel_clock": frac(kn_ms), "launches_per_spmv": int(np.median(kl)),
            "us_cold": None if c_ms is None else round(c_ms * 1e3, 2), "frac_cold": None if c_ms is None else frac(c_ms),
            "cached_share_of_frac": None if c_ms is None else round(1.0 - reset_ms / c_ms, 4),
            "col16": info.get("col16", -1),  # ints per chunk record if the kernel reads the plan's 16-bit column encoding, 0 = the caller's colindex
            "gflops": round(2.0 * nnz / (reset_ms * 1e-3) / 1e9, 1),
            "plan": [info.get(k, -1) for k in ("stream_policy", "adaptive_family", "flat_fixup")], "kernel": info.get("last_kernel"),
            "settled": bool(info.get("settled"))}


def _leg_vectors(torch, device, m, n):
    gen = torch.Generator(device=device)
    gen.manual_seed(1234)
    return (torch.rand(n, generator=gen, device=device, dtype=torch.float64) * 2 - 1,
            torch.rand(m, generator=gen, device=device, dtype=torch.float64) * 2 - 1)


def leg_sweep(torch, device, name, A=None, light=False):
    """One stand-in of the configs[2] sweep: flat (the strategy BASELINE names), adaptive, and (not `light`) flat's tile kernel alone
    and the opt-in 16-bit column stream."""
    import spmv_acc_amd
    from spmv_acc_amd import synth

    if A is None:
        A = synth.sweep_standin_torch(name, device=device)
    x, y0 = _leg_vectors(torch, device, A[0], A[1])
    iters = 200 if A[2] < 20_000_000 else 60
    row = {"rows": A[0], "nnz": A[2]}
    for strat in ("flat", "adaptive"):
        row[strat] = timed_leg(torch, strat, A, x, y0, iters)
    if light:
        spmv_acc_amd.release_plans(A[3])
        return row
    lib = spmv_acc_amd.load_library()
    # `flat` as shipped runs, on balanced rows, the faster of its tile kernel and the row-block kernel (timed once per matrix, tunable
    # flat_rowblock); this leg pins the TILE kernel, so that the line shows both
    spmv_acc_amd.release_plans(A[3])
    lib.spmv_acc_set_tunable(b"flat_rowblock", 0)
    try:
        row["flat_tile_kernel"] = timed_leg(torch, "flat", A, x, y0, iters)
    finally:
        lib.spmv_acc_set_tunable(b"flat_rowblock", -1)
    # the 16-bit column encoding is the library's timed choice since round 6 (tunable col16 = -1; `col16` in every leg says whether the kernel reads it):
    # this leg pins the caller's colindex (col16 = 0), so that the record shows what the encoding is worth where a plan keeps it
    spmv_acc_amd.release_plans(A[3])
    lib.spmv_acc_set_tunable(b"col16", 0)
    try:
        row["adaptive_colindex_only"] = timed_leg(torch, "adaptive", A, x, y0, iters)
    finally:
        lib.spmv_acc_set_tunable(b"col16", -1)
    spmv_acc_amd.release_plans(A[3])
    return row


def leg_rmat25(torch, device):
    """BASELINE configs[3]: R-MAT scale 25 under line_enhance -- the default path (since round 6: the slab-major copy, built inside spmv_acc_prepare where
    it times faster than the run-list passes and 3 x 12 B per non-zero are free), the run-list passes alone (col_slabs = 0: what rounds 3-5 ran by
    default), and the one-kernel path both are timed against."""
    import spmv_acc_amd
    from spmv_acc_amd import synth

    A = synth.rmat_torch(25, device=device, seed=0xC4)
    x, y0 = _leg_vectors(torch, device, A[0], A[1])
    out = {"workload": "R-MAT scale 25, edge factor 16 (BASELINE configs[3])", "rows": A[0], "nnz": A[2],
           "line_enhance": timed_leg(torch, "line_enhance", A, x, y0, iters=10, warm=3)}
    info = spmv_acc_amd.query_plan(A[3], A[0]) or {}
    kernel = info.get("last_kernel")
    out["path"] = ("slab-major copy of the matrix in %d column slabs, built lazily and timed against the column-slab passes over run lists (k_slab.hip; values "
                   "guarded by 65,536 samples per call)" % info.get("slab_passes", 0) if kernel == "col_slabs" else
                   "column-slab passes over run lists, no copy of the matrix (k_segment.hip; timed against the row-block-plus kernel at plan time)"
                   if info.get("slab_passes") else "row-block-plus kernel")
    spmv_acc_amd.release_plans(A[3])
    lib = spmv_acc_amd.load_library()
    # the run-list passes alone (no copy of the matrix: the default of rounds 3-5)
    lib.spmv_acc_set_tunable(b"col_slabs", 0)
    try:
        out["line_enhance_slab_passes_only"] = timed_leg(torch, "line_enhance", A, x, y0, iters=10, warm=3)
    finally:
        lib.spmv_acc_set_tunable(b"col_slabs", -1)
        spmv_acc_amd.release_plans(A[3])
    # the same strategy with neither: the one-kernel path of rounds 1-2 (gather hints), for comparison
    lib.spmv_acc_set_tunable(b"slab_segments", 0)
    try:
        out["line_enhance_without_slab_passes"] = timed_leg(torch, "line_enhance", A, x, y0, iters=10, warm=3)
    finally:
        lib.spmv_acc_set_tunable(b"slab_segments", -1)
        spmv_acc_amd.release_plans(A[3])
    return out


def leg_banded_shard(torch, device):
    """BASELINE configs[4] on one GPU: rank 3's 32 M-row shard of the 256 M-row banded matrix, beta = 0."""
    import spmv_acc_amd
    from spmv_acc_amd import synth

    rows, total = 32_000_000, 256_000_000
    rp, ci, v = synth.banded_torch(rows, first_row=3 * rows, total_rows=total, device=device)
    A = (rows, total, int(rp[-1].item()), rp, ci, v)
    x, y0 = _leg_vectors(torch, device, rows, total)
    out = {"workload": "rank 3's 32 M-row shard of the 256 M-row banded matrix (BASELINE configs[4]), beta = 0",
           "rows": rows, "nnz": A[2], "adaptive": timed_leg(torch, "adaptive", A, x, y0, iters=30, warm=5, beta=0.0, cols_touched=rows + 7),
           "algorithmic_bytes_note": "x counted over the rows + 7 columns the shard references, not over all 256 M"}
    spmv_acc_amd.release_plans(rp)
    return out


def leg_child(args):
    """`python bench.py --leg-child <spec>`: ONE leg in a process of its own, its JSON on stdout (last line)."""
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)  # libraries' chatter to stderr, as in main()
    import torch

    import spmv_acc_amd

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    spmv_acc_amd.load_library()
    spec = args.leg_child
    if spec.startswith("sweep:"):
        out = leg_sweep(torch, device, spec[6:])
    elif spec == "rmat25":
        out = leg_rmat25(torch, device)
    elif spec == "banded_shard":
        out = leg_banded_shard(torch, device)
    elif spec == "sweep_in_one_process":
        out = leg_sweep_in_one_process(torch, device)
    else:
        raise SystemExit(f"unknown leg {spec}")
    os.write(json_fd, (json.dumps(out) + "\n").encode())


def leg_sweep_in_one_process(torch, device, resident=None, progress=None):
    """The 11 other sweep stand-ins measured one after the other in ONE process that also holds the headline matrix (what a solver holding several
    matrices sees; placement moves the 65 M-non-zero stand-ins by 2-4 %).  A child process of its own since round 4's end, so that the parent's
    rocprofv3 kernel summary holds the headline matrix's launches only; `resident` = the headline matrix when the parent has to do it itself."""
    from spmv_acc_amd import synth

    keep = resident if resident is not None else synth.hardesty3_like_torch(device=device, seed=0xC2)
    out = {}
    for name in synth.SWEEP_NAMES:
        if name == "Hardesty3":
            continue
        r = leg_sweep(torch, device, name, light=True)
        out[name] = {s: r[s] for s in ("flat", "adaptive")}
        torch.cuda.empty_cache()
        if progress:
            progress(f"sweep (in one process) {name}: flat {out[name]['flat']['frac']}, adaptive {out[name]['adaptive']['frac']}")
    del keep
    return out


def run_leg_child(spec):
    """One leg in a child process (never an exec: this process holds the GPU).  Returns the leg's dict, or None if the child failed."""
    import subprocess

    env = dict(os.environ)
    env.pop("SPMV_ACC_TUNE_LOG", None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg-child", spec], env=env, capture_output=True, text=True, timeout=900)
    except Exception as ex:  # noqa: BLE001
        print(f"[bench legs] {spec}: child did not finish ({ex!r}); measuring in this process", file=sys.stderr, flush=True)
        return None
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        print(f"[bench legs] {spec}: child failed (exit {r.returncode}): {r.stderr[-400:]!r}; measuring in this process", file=sys.stderr, flush=True)
        return None
    return json.loads(lines[-1])


def extra_legs(torch, device, headline, in_process=False):
    """BASELINE configs[2..4] measured live in the same run (rank 0, N = 1): the 12-matrix sweep under flat (the strategy
    BASELINE names) and adaptive, R-MAT scale 25 under line_enhance, one 32 M-row banded shard.  Keys are additions to the
    one JSON line; the headline value is untouched.

    Every matrix is measured in a CHILD PROCESS of its own, as the reference's batch does (examples/large-data-set-batch.sh runs spmv-cli once per
    matrix file): a matrix measured in a process that already holds others lands on whatever physical pages are left, which is worth 2-4 % on the
    65 M-non-zero stand-ins (profiles/r03_leg_spread_probe.txt: Bump_2911-sized 149 us alone, 152-156 us after the headline matrix and the earlier
    stand-ins).  The Hardesty3 entry of the sweep is the headline's own matrix and stays here.  `--legs-in-process` measures everything here."""
    import spmv_acc_amd
    from spmv_acc_amd import synth

    def progress(msg):  # stderr: the one JSON line on stdout stays alone (and a long run is seen to be alive)
        print(f"[bench legs] {msg}", file=sys.stderr, flush=True)

    def leg(spec, here):
        got = None if in_process else run_leg_child(spec)
        if got is None:
            got = here()
            torch.cuda.empty_cache()
        return got

    out = {"legs_measured": "in this process" if in_process else "one child process per matrix (examples/large-data-set-batch.sh: one spmv-cli run per file)"}
    sweep = {}
    for name in synth.SWEEP_NAMES:
        if name == "Hardesty3":
            sweep[name] = leg_sweep(torch, device, name, A=headline)
        else:
            sweep[name] = leg(f"sweep:{name}", lambda name=name: leg_sweep(torch, device, name))
        progress(f"sweep {name}: flat {sweep[name]['flat']['us']} us ({sweep[name]['flat']['frac']}; tile kernel alone {sweep[name]['flat_tile_kernel']['us']} us), "
                 f"adaptive {sweep[name]['adaptive']['us']} us ({sweep[name]['adaptive']['frac']}), "
                 f"adaptive on colindex only {sweep[name]['adaptive_colindex_only']['us']} us ({sweep[name]['adaptive_colindex_only']['frac']}); "
                 f"col16 flat / adaptive {sweep[name]['flat']['col16']} / {sweep[name]['adaptive']['col16']}; cold {sweep[name]['flat']['frac_cold']} / {sweep[name]['adaptive']['frac_cold']}")
    out["sweep"] = sweep
    # The same 12 stand-ins once more with every matrix measured IN THIS PROCESS, one after the other (what a solver holding several
    # matrices sees): the >= 0.70 count is quoted for both regimes, because placement moves the 65 M-non-zero stand-ins by 2-4 %.
    inproc = None
    if not in_process:
        inproc = leg("sweep_in_one_process", lambda: leg_sweep_in_one_process(torch, device, resident=headline, progress=progress))
        inproc["Hardesty3"] = {s: sweep["Hardesty3"][s] for s in ("flat", "adaptive")}
        progress("sweep (in one process): " + ", ".join(f"{k} {v['flat']['frac']} / {v['adaptive']['frac']}" for k, v in inproc.items()))
        out["sweep_in_process"] = inproc
    out["sweep_summary"] = {
        s: {"protocol": "per-launch, y reset (benchmark/csr_spmv.hpp:66-74); *_back_to_back beside it",
            "ge_0.70": sum(1 for r in sweep.values() if r[s]["frac"] >= 0.70),
            "min_frac": min(r[s]["frac"] for r in sweep.values()),
            "median_frac": float(np.median([r[s]["frac"] for r in sweep.values()])),
            "ge_0.70_back_to_back": sum(1 for r in sweep.values() if r[s]["frac_back_to_back"] >= 0.70),
            "ge_0.70_kernel_clock": sum(1 for r in sweep.values() if r[s]["frac_kernel_clock"] >= 0.70),
            "median_frac_kernel_clock": float(np.median([r[s]["frac_kernel_clock"] for r in sweep.values()])),
            "median_frac_back_to_back": float(np.median([r[s]["frac_back_to_back"] for r in sweep.values()])),
            "ge_0.70_cold": sum(1 for r in sweep.values() if (r[s].get("frac_cold") or 0.0) >= 0.70),
            "median_frac_cold": float(np.median([r[s].get("frac_cold") or 0.0 for r in sweep.values()])),
            "median_cached_share_of_frac": float(np.median([r[s].get("cached_share_of_frac") or 0.0 for r in sweep.values()])),
            "stand_ins_on_16_bit_columns": sum(1 for r in sweep.values() if (r[s].get("col16") or 0) > 0)}
        for s in ("flat", "adaptive", "flat_tile_kernel", "adaptive_colindex_only")}
    for s in ("flat", "adaptive"):
        rows = (inproc or sweep).values()
        out["sweep_summary"][s]["ge_0.70_in_process"] = sum(1 for r in rows if r[s]["frac"] >= 0.70)
        out["sweep_summary"][s]["median_frac_in_process"] = float(np.median([r[s]["frac"] for r in rows]))
    out["rmat25"] = leg("rmat25", lambda: leg_rmat25(torch, device))
    progress(f"rmat25: {out['rmat25']['line_enhance']} ({out['rmat25']['path']})")
    progress(f"rmat25, run-list passes only (no copy): {out['rmat25']['line_enhance_slab_passes_only']}")
    progress(f"rmat25 without the slab passes: {out['rmat25']['line_enhance_without_slab_passes']}")
    out["banded_shard"] = leg("banded_shard", lambda: leg_banded_shard(torch, device))
    progress(f"banded shard: {out['banded_shard']['adaptive']}")
    return out


def sharded_leg(torch, dist, args, W, x, y0, alpha, beta, rank, world, device, backend, force_dist, steps, warmup, bounds=None, time_left=None):
    """N > 1: one RowShardedSpmv over this rank's shard -- local SpMV on the engine's own stream, ONE exchange of the y slices
    per step (RCCL allgather or the point-to-point fan-out, whichever the timing on this communicator prefers).  Returns the
    max-over-ranks wall time of `steps` steps (barrier + synchronize on both sides) and the SpMV-only launch time."""
    import spmv_acc_amd
    from spmv_acc_amd.dist import RowShardedSpmv

    m, n, nnz = W["m"], W["n"], W["nnz"]
    strat = W["strategy"]
    extra = {}
    if bounds is None:
        bounds = np.arange(world + 1, dtype=np.int64) * m  # every rank owns m rows of the (world*m) x n matrix
    eng = RowShardedSpmv(rank, world, bounds, W["rp"], W["ci"], W["v"], n, device, strategy=strat,
                         always_collective=force_dist,
                         exchange="allgather" if args.exchange == "auto" else args.exchange)
    if args.exchange == "auto" and backend == "nccl":  # (gloo rehearsals: no point-to-point on GPU tensors)
        try:
            extra["exchange_ms"] = {k: round(v, 4) for k, v in eng.tune_exchange().items()}
        except Exception as ex:  # noqa: BLE001 -- a backend without grouped point-to-point: keep the collective
            extra["exchange_tune_error"] = repr(ex)[:200]
            eng.exchange = "allgather"
    extra["exchange"] = eng.exchange

    def sync_all():
        dist.barrier()
        torch.cuda.synchronize()

    eng.prepare(beta, x)  # every per-matrix timing up front (spmv_acc_prepare): none is left to fall into the timed steps
    eng.set_y(y0)  # like the N = 1 leg, y is iterated in place (no per-step reset inside the timed region)
    for _ in range(max(warmup, 1)):
        eng.step(alpha, beta, x, overlap=not args.no_overlap)
    eng.wait()

    def region():
        for _ in range(steps):
            eng.step(alpha, beta, x, overlap=not args.no_overlap)
        eng.wait()

    def max_over_ranks(values):
        t = torch.tensor(values, dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return t.tolist()

    # the K-step region `reps` times (each bracketed by barrier + synchronize, MAX over ranks per repetition), median; a first repetition that
    # already takes long (a slow exchange) is not repeated as often: the run must end inside the driver's timeout
    wall1, _ = median_region(region, 1, sync_all, max_over_ranks)
    reps = REGION_REPS if wall1 * REGION_REPS < 30.0 else 3
    wall, walls = median_region(region, reps, sync_all, max_over_ranks)
    extra["region_reps"] = reps
    extra["ms_per_step_wall_all"] = [round(w / steps * 1e3, 6) for w in walls]
    # SpMV-only (no exchange) for the same shard, per-launch hipEvents
    y = y0.clone()
    ms = spmv_acc_amd.time_spmv(strat, min(steps, 50), alpha, beta, m, n, nnz, W["rp"], W["ci"], W["v"], x, y)
    ev_ms = float(np.mean(ms))
    ev_ms_max = max_over_ranks([ev_ms])[0]
    extra["spmv_only_gflops_per_gpu"] = round(2.0 * nnz / (ev_ms * 1e-3) / 1e9, 3)
    extra["spmv_only_ms_max_over_ranks"] = round(ev_ms_max, 6)
    extra["spmv_plus_exchange_ms_per_step"] = round(wall / steps * 1e3, 6)
    extra["spmv_plus_exchange_gflops_total"] = round(2.0 * nnz * world * steps / wall / 1e9, 3)
    extra["allgather_bytes_per_rank_per_step"] = 8 * eng.pad * (world - 1)
    # The DEPENDENT step: x_{k+1} = f(gathered y_k) solvers cannot start step k+1 before the exchange of step k has ended, so the
    # cross-step overlap above (legal here only because x is fixed) does not exist for them.  What does: cutting the local rows
    # into C chunks and sending chunk c while chunk c+1 computes (RowShardedSpmv.step(pipeline=C)).  Timed with every step
    # waiting for its exchange; C = 1 is the serial kernel + exchange.
    # (a collective decision: `time_left()` is the MINIMUM over ranks of what is left of the run's allowance -- every rank skips or none does)
    if backend == "nccl" and (time_left is None or time_left() > 60.0):
        try:
            dep = eng.tune_pipeline(alpha, beta, x, candidates=(1, 2, 4, 8), warm=2, iters=max(3, min(steps, 10)))
            extra["dependent_step_ms_by_pipeline"] = {str(k): round(v, 6) for k, v in dep.items()}
            extra["pipeline_best"] = eng.pipeline
        except Exception as ex:  # noqa: BLE001
            extra["pipeline_tune_error"] = repr(ex)[:200]
    del eng
    return wall, ev_ms, extra


def rccl_debug_summary(rank):
    """What RCCL itself logged about this job's communicator (NCCL_DEBUG=INFO into a per-rank file): the rank count it
    saw and, where the build prints them, the algorithm / protocol lines -- so that 'RCCL ran over N ranks' can be checked
    from the bench line."""
    import re

    path = os.environ.get("NCCL_DEBUG_FILE", "").replace("%h", os.uname().nodename).replace("%p", str(os.getpid()))
    out = {"nranks_seen": None, "debug_file": path, "debug_file_bytes": None, "lines": []}
    try:
        text = open(path, errors="replace").read()
    except OSError as ex:
        out["debug_file_error"] = repr(ex)[:120]
        return out
    out["debug_file_bytes"] = len(text)
    m = re.findall(r"nranks (\d+)", text)
    if m:
        out["nranks_seen"] = int(m[-1])
    picked = [ln for ln in text.splitlines()
              if re.search(r"Init COMPLETE|Algo|algorithm|Connected all|Channel 00[ /:]|via P2P|via SHM|via NET|nranks", ln)]
    if not picked:
        picked = [ln for ln in text.splitlines() if "NCCL" in ln]
    out["lines"] = [re.sub(r"^.*NCCL INFO ", "", ln)[:160] for ln in picked[:8]]
    return out


def copy_ceiling_gbs(torch, device):
    """Device streaming-copy ceiling measured in the same run (1 GiB, the kernels' 16-B nt access shape)."""
    import spmv_acc_amd

    n = 1 << 27
    a = torch.empty(n, dtype=torch.float64, device=device).normal_()
    b = torch.empty_like(a)
    lib = spmv_acc_amd.load_library()
    best = spmv_acc_amd.copy_ceiling_gbs(b, a, reps=5)  # (the entry times the non-temporal and the default-policy copy and returns the faster)
    return best


LINE_BUDGET = 6000  # the driver keeps an 8,001-byte tail of stdout: the ONE line must fit it with room to spare


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full):
    """The ONE stdout line: the contract's keys, `roofline`, `cpu_baseline` and one short figure per extra measurement -- as the
    reference's harness prints one short `PERFORMANCE,` row per measurement (benchmark/utils/statistics_logger.cpp:11-31).
    Everything else (prose notes, the third timing column, opt-in legs, launch floor, sensitivity, RCCL log lines) lives in
    bench_full.json beside this script and on stderr."""
    line = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                        "vs_baseline", "dtype", "data", "config", "region_reps", "ms_per_step_events"))
    r = full["roofline"]
    line["roofline"] = _pick(r, ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_lower_bound",
                                 "algorithmic_bytes_per_launch", "launch_ms_mean", "frac_cold", "cached_share_of_frac"))
    line["roofline"]["definition"] = _pick(r.get("definition", {}), ("version", "frac", "legs_and_gates"))
    line["roofline"]["back_to_back"] = _pick(r.get("back_to_back", {}), ("frac",))
    line["roofline"]["per_launch_protocol"] = _pick(r.get("per_launch_protocol", {}), ("frac", "launch_ms_median"))
    line["roofline"]["kernel_clock_reset_protocol"] = _pick(r.get("kernel_clock_reset_protocol", {}), ("frac", "launch_ms_median"))
    if "frac_of_copy_ceiling" in r:
        line["roofline"]["frac_of_copy_ceiling"] = r["frac_of_copy_ceiling"]
    cb = full.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else _pick(cb, ("value", "unit", "cores", "kind", "sample", "cpu_model", "value_1thread", "value_median_per_round",
                                                                  "stream_triad_gbs", "frac_of_stream_triad", "value_best", "placement"))
    line.update(_pick(full, ("copy_ceiling_gbs", "plan", "first_call_ms", "settle_rest_ms")))
    if "sweep" in full:  # configs[2]: name -> [flat frac, adaptive frac] under the per-launch protocol
        line["sweep"] = {k: [row["flat"]["frac"], row["adaptive"]["frac"]] for k, row in full["sweep"].items()}
        line["sweep_summary"] = {s: _pick(v, ("ge_0.70", "ge_0.70_in_process", "median_frac", "min_frac", "median_frac_in_process", "ge_0.70_kernel_clock",
                                              "median_frac_kernel_clock", "ge_0.70_cold", "median_frac_cold", "stand_ins_on_16_bit_columns"))
                                 for s, v in full["sweep_summary"].items() if s in ("flat", "adaptive")}
        # (`flat` as shipped runs the row-block kernel on balanced rows where the plan-time timing prefers it: flat's own tile kernel ALONE beside it)
        if "flat_tile_kernel" in full["sweep_summary"]:
            line["sweep_summary"]["flat"]["ge_0.70_tile_kernel_alone"] = full["sweep_summary"]["flat_tile_kernel"]["ge_0.70"]
    if "rmat25" in full:  # configs[3]
        le = full["rmat25"]["line_enhance"]
        line["rmat25"] = {"us": le["us"], "frac": le["frac"], "nnz": full["rmat25"]["nnz"], "frac_cold": le.get("frac_cold")}
    if "banded_shard" in full:  # configs[4], one shard
        ad = full["banded_shard"]["adaptive"]
        line["banded_shard"] = {"us": ad["us"], "frac": ad["frac"], "frac_cold": ad.get("frac_cold")}
    if "sensitivity" in full:
        line["no_far_columns_frac"] = full["sensitivity"]["frac"]
    # N > 1
    line.update(_pick(full, ("exchange", "spmv_only_gflops_per_gpu", "spmv_only_ms_max_over_ranks", "spmv_plus_exchange_ms_per_step",
                             "spmv_plus_exchange_gflops_total", "allgather_bytes_per_rank_per_step", "dependent_step_ms_by_pipeline",
                             "pipeline_best", "exchanged_bytes_per_rank_per_step")))
    if "rccl" in full:
        line["rccl"] = {"ranks": full["rccl"].get("nranks_seen"), "ranks_ok": full["rccl"].get("nranks_seen") == full.get("n_gpus")}
    if "banded" in full:
        b = full["banded"]
        line["banded"] = _pick(b, ("rows_per_gpu", "nnz_per_gpu", "steps", "exchange", "spmv_only_frac_of_hbm_peak", "spmv_only_gflops_per_gpu",
                                   "spmv_plus_exchange_ms_per_step", "spmv_plus_exchange_gflops_total", "dependent_step_ms_by_pipeline"))
        if "halo_exchange" in b:
            line["banded"]["halo_exchange"] = _pick(b["halo_exchange"], ("ms_per_step", "gflops_total"))
    if "strong_scaling" in full:
        line["strong_scaling"] = _pick(full["strong_scaling"], ("gflops_total", "ms_per_step", "exchange", "spmv_only_ms_max_over_ranks"))
    line["details"] = "bench_full.json"
    text = json.dumps(line, separators=(",", ":"))
    for drop in ("strong_scaling", "banded", "sweep", "plan", "copy_ceiling_gbs"):  # never reached today; the line must fit whatever is added later
        if len(text) < LINE_BUDGET:
            break
        line.pop(drop, None)
        text = json.dumps(line, separators=(",", ":"))
    return text


def write_full(full):
    """The whole record: bench_full.json beside the script (best effort: a read-only checkout only loses the file) and stderr."""
    text = json.dumps(full, indent=1)
    try:
        with open(os.path.join(ROOT, "bench_full.json"), "w") as f:
            f.write(text + "\n")
    except OSError as ex:
        print(f"[bench] bench_full.json not written: {ex!r}", file=sys.stderr)
    print("[bench full record]\n" + text, file=sys.stderr, flush=True)


def self_launch(args):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks as CHILD processes and relay rank 0's
    line.  This parent never imports torch or the library -- nothing here touches the GPU, and nothing is exec'ed from a process
    that has: the children are started with subprocess (torch.distributed.run's elastic agent, itself GPU-free, spawns the
    ranks).  Exit status = the children's."""
    import socket
    import subprocess

    with socket.socket() as s:  # a free rendezvous port (the driver may run several sizes back to back)
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env["SPMV_ACC_BENCH_CHILD"] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:  # rank 0 writes exactly one JSON line; anything else a child prints on stdout goes to stderr
        if out.startswith("{") and line is None:
            line = out
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        sys.stdout.write(line)
        sys.stdout.flush()
    if rc == 0 and line is None:
        print("[bench] the ranks exited without a JSON line", file=sys.stderr)
        rc = 1
    raise SystemExit(rc)


def dry_run(args):
    """SPMV_ACC_BENCH_DRYRUN=1 (CPU test of the N > 1 contract, no GPU): the ranks rendezvous over gloo and run everything about an N > 1 run
    except the GPU work -- the repeated K-step regions with their barriers and the MAX-over-ranks of every repetition (median_region), the
    collective time-budget decision -- and rank 0 prints the ONE line through compact_line() from a record that carries EVERY key a real
    N > 1 run produces (values are placeholders of realistic width), so that the line's length at N = 8 is tested on CPU."""
    import torch
    import torch.distributed as dist

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([1.0 + rank])
    dist.barrier()
    dist.all_reduce(t)

    def max_over_ranks(values):
        v = torch.tensor(values, dtype=torch.float64)
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return v.tolist()

    def time_left():
        v = torch.tensor([budget_left(float(os.environ.get("SPMV_ACC_BENCH_BUDGET_S", "420")))], dtype=torch.float64)
        dist.all_reduce(v, op=dist.ReduceOp.MIN)
        return float(v.item())

    # rank r's fake step takes (1 + r) ms: the MAX over ranks of every repetition is the slowest rank's
    wall, walls = median_region(lambda: time.sleep(1e-3 * (1 + rank) * args.steps), 3, dist.barrier, max_over_ranks)
    left = time_left()
    if rank == 0:
        leg = {"exchange": "allgather", "exchange_ms": {"allgather": 1.2345, "p2p": 1.3456}, "spmv_only_gflops_per_gpu": 512.345,
               "spmv_only_ms_max_over_ranks": 0.157912, "spmv_plus_exchange_ms_per_step": 1.234567, "spmv_plus_exchange_gflops_total": 4321.123,
               "allgather_bytes_per_rank_per_step": 8 * 8217820 * (world - 1), "region_reps": 3,
               "ms_per_step_wall_all": [round(w / args.steps * 1e3, 6) for w in walls],
               "dependent_step_ms_by_pipeline": {"1": 1.234567, "2": 1.123456, "4": 1.012345, "8": 1.001234}, "pipeline_best": 4}
        full = {"metric": "dry run (launch contract only)", "value": 0.0, "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(wall / args.steps * 1e3, 6), "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": "Hardesty3-like (SuiteSparse Hardesty3 dims, synthetic stand-in)", "rows_per_gpu": 8217820, "cols": 7591564,
                           "nnz_per_gpu": 40451632, "strategy": "adaptive", "alpha": 1.0, "beta": 1.0, "scale": 1.0,
                           "parallelism": f"row-range shard x{world} + allgather(y) over gloo (dry run)"},
                "roofline": {"bound": "hbm", "achieved": 4500.12, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.5625, "traffic": None,
                             "traffic_lower_bound": None, "algorithmic_bytes_per_launch": 710508500, "launch_ms_mean": 0.157912,
                             "back_to_back": {"launch_ms_mean": 0.157912, "achieved": 4500.12, "frac": 0.5625}, "frac_of_copy_ceiling": 0.7012},
                "cpu_baseline": None, "copy_ceiling_gbs": 6400.1,
                "plan": {"stream_policy": 0, "adaptive_family": 0, "flat_fixup": -1, "plus_blocks": -1, "flat_tiles": -1, "slab_passes": 0},
                "rccl": {"nranks_seen": world, "debug_file": "/tmp/x", "debug_file_bytes": 12345, "lines": ["Init COMPLETE " + "w" * 140] * 8},
                "banded": dict(leg, workload="banded " + "x" * 200, rows_per_gpu=32_000_000, nnz_per_gpu=255_999_993, steps=50,
                               spmv_only_frac_of_hbm_peak=0.7123,
                               halo_exchange={"what": "y" * 120, "ms_per_step": 0.712345, "gflops_total": 5432.123,
                                              "exchanged_bytes_per_rank_per_step": 112, "ghost_columns": 7}),
                "strong_scaling": dict(leg, workload="z" * 100, rows_this_rank=1027227, nnz_this_rank=5056454, steps=100,
                                       gflops_total=3210.123, ms_per_step=0.025123),
                "dry_run": True, "ranks_sum": float(t.item()), "time_left_s": round(left, 1),
                "launched_by": "self" if os.environ.get("SPMV_ACC_BENCH_CHILD") == "1" else "external launcher"}
        full.update(leg)
        line = json.loads(compact_line(full))
        line.update({k: full[k] for k in ("dry_run", "ranks_sum", "launched_by", "time_left_s")})
        print(json.dumps(line, separators=(",", ":")), flush=True)
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("SPMV_ACC_BENCH_CHILD") != "1":
        self_launch(args)  # does not return
    if os.environ.get("SPMV_ACC_BENCH_DRYRUN", "0") == "1":
        dry_run(args)
        return
    if args.leg_child:
        leg_child(args)
        return
    if args.cpu_baseline_child:
        cpu_baseline_child(args)
        return
    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's version banner with NCCL_DEBUG set, loader
    # notices): from here on file descriptor 1 is stderr, and the JSON line is written to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("SPMV_ACC_BENCH_FORCE_DIST", "0") == "1":
        # RCCL's own account of the communicator goes to a per-rank file and is quoted in the JSON line (rccl_debug_summary).
        # Set BEFORE torch (and with it librccl) is loaded: measured on this image, the same variables set after `import torch`
        # turn the version banner on but never create the file.
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("SPMV_ACC_BENCH_BACKEND", "nccl") == "nccl":
            # (assigned, not setdefault: the GPU boxes of this pool export NCCL_DEBUG=VERSION, which prints the banner and nothing else)
            os.environ["NCCL_DEBUG"] = "INFO"
            os.environ.setdefault("NCCL_DEBUG_FILE", f"/tmp/spmv_acc_bench_rccl_{os.getpid()}_rank{os.environ.get('RANK', '0')}.log")
    import torch
    import torch.distributed as dist

    import spmv_acc_amd
    from spmv_acc_amd import synth
    from spmv_acc_amd.dist import RowShardedSpmv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    # rehearsal only: SPMV_ACC_BENCH_BACKEND=gloo lets several ranks share one GPU (RCCL needs one GPU per rank)
    backend = os.environ.get("SPMV_ACC_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    spmv_acc_amd.load_library()
    # rehearsal only: SPMV_ACC_BENCH_FORCE_DIST=1 runs the N > 1 code path (process group, RowShardedSpmv, allgather,
    # max-over-ranks) on however many ranks there are, including one -- the only RCCL run a one-GPU box allows
    force_dist = os.environ.get("SPMV_ACC_BENCH_FORCE_DIST", "0") == "1"
    dist_leg = world > 1 or force_dist
    if dist_leg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    W = build_workload(args, torch, device, rank)
    m, n, nnz = W["m"], W["n"], W["nnz"]
    gen = torch.Generator(device=device)
    gen.manual_seed(1234)  # x is replicated: same seed on every rank
    x = torch.rand(n, generator=gen, device=device, dtype=torch.float64) * 2 - 1
    y0 = torch.rand(m, generator=gen, device=device, dtype=torch.float64) * 2 - 1
    y = y0.clone()
    strat = W["strategy"]
    alpha = beta = 1.0  # cli/main.cpp:95-96, benchmark/main.cpp:101-102

    def sync_all():
        if dist_leg:
            dist.barrier()
        torch.cuda.synchronize()

    out_extra = {}
    if not dist_leg:
        # ---- warm-up (builds the plan: nnz / samples / break points are fetched once here) ----
        # (one throw-away call on a 1024-row matrix first: loading the kernels' code objects is the process's cost, not this matrix's)
        # (under `deterministic`: ONE launch, no trial launches of a kernel instance the headline matrix may share -- they would sit in the rocprofv3
        # summary of that kernel as twenty 3-us dispatches)
        tiny = torch.arange(1025, dtype=torch.int32, device=device)
        ty = torch.zeros(1024, dtype=torch.float64, device=device)
        _lib = spmv_acc_amd.load_library()
        _det = _lib.spmv_acc_get_tunable(b"deterministic")
        _lib.spmv_acc_set_tunable(b"deterministic", 1)
        spmv_acc_amd.csr_spmv(1.0, 1.0, 1024, 1024, 1024, tiny, tiny[:1024].contiguous(), torch.ones(1024, dtype=torch.float64, device=device),
                              torch.ones(1024, dtype=torch.float64, device=device), ty, strategy=strat)
        spmv_acc_amd.release_plans(tiny)
        _lib.spmv_acc_set_tunable(b"deterministic", _det)
        del tiny, ty
        torch.cuda.synchronize()
        tf = time.perf_counter()
        for i in range(max(args.warmup, 1)):
            spmv_acc_amd.csr_spmv(alpha, beta, m, n, nnz, W["rp"], W["ci"], W["v"], x, y, strategy=strat)
            if i == 0:
                # The first call builds the plan: structural passes + the per-matrix timings its tuning budget allows (tunable first_call_budget,
                # 20 SpMV-equivalents; the rest is finished lazily by the following calls).  Reported, never inside the timed region -- and so that
                # none of the lazily finished timings falls into it either, everything still open is settled here, up front (spmv_acc_prepare).
                torch.cuda.synchronize()
                out_extra["first_call_ms"] = round((time.perf_counter() - tf) * 1e3, 3)
                out_extra["first_call_host_prepare_ms"] = round(spmv_acc_amd.load_library().spmv_acc_last_prepare_us() * 1e-3, 3)
                out_extra["settle_rest_ms"] = round(spmv_acc_amd.prepare(m, n, nnz, W["rp"], W["ci"], W["v"], x, strategy=strat, beta=beta), 3)
        torch.cuda.synchronize()
        y.copy_(y0)
        # timed region: exactly K back-to-back launches between one hipEvent pair on the library stream and nothing else (the plan is settled,
        # the events exist, arguments are converted once: spmv_acc_time_spmv_region); one untimed region first, then REGION_REPS timed ones
        region = spmv_acc_amd.time_spmv_region(strat, args.steps, alpha, beta, m, n, nnz, W["rp"], W["ci"], W["v"], x, y)
        region()
        region_event_ms = []
        wall, walls = median_region(lambda: region_event_ms.append(region()), REGION_REPS, sync_all)
        b2b_ms = float(np.median(region_event_ms)) / args.steps
        out_extra["region_reps"] = REGION_REPS
        out_extra["ms_per_step_events"] = round(b2b_ms, 6)
        out_extra["ms_per_step_wall_all"] = [round(w / args.steps * 1e3, 6) for w in walls]
        out_extra["ms_per_step_events_all"] = [round(e / args.steps, 6) for e in region_event_ms]
        # the reference harness's protocol (csr_spmv.hpp:66-74): y reset by a device copy before every launch, one event pair
        # per launch, median -- roofline.per_launch_protocol (rounds 1-4 quoted roofline.frac on it; since round 5 roofline.frac is the timed
        # region's back-to-back figure: ROOFLINE_DEFINITION above, frozen)
        ms = spmv_acc_amd.time_spmv(strat, max(20, min(args.steps, 50)), alpha, beta, m, n, nnz, W["rp"], W["ci"], W["v"], x, y, y0=y0)
        ev_ms = float(np.median(ms))
        # the same protocol with the library's kernel clock on: each launch's own start / stop timestamps
        ev2, kn_ms, kn_launches = spmv_acc_amd.time_spmv_kernels(strat, max(20, min(args.steps, 50)), alpha, beta, m, n, nnz, W["rp"], W["ci"], W["v"], x, y, y0=y0)
        kernel_ms = float(np.median(kn_ms))
        out_extra["kernel_clock_ms_median"] = round(kernel_ms, 6)
        out_extra["kernel_clock_launches_per_step"] = int(np.median(kn_launches))
        out_extra["per_launch_reset_ms_median_under_kernel_clock"] = round(float(np.median(ev2)), 6)
        out_extra["per_launch_reset_ms_median"] = round(ev_ms, 6)
        out_extra["per_launch_reset_ms_min"] = round(float(np.min(ms)), 6)
        out_extra["back_to_back_ms_mean"] = round(b2b_ms, 6)
        out_extra["per_launch_reset_ms_median_events_without_system_fence"] = round(
            nofence_ms(strat, (m, n, nnz, W["rp"], W["ci"], W["v"]), x, y, y0, 20, alpha, beta), 6)
        _cold = cold_ms(torch, strat, (m, n, nnz, W["rp"], W["ci"], W["v"]), x, y, y0, 20, alpha, beta)
        if _cold is not None:
            out_extra["per_launch_cold_ms_median"] = round(_cold, 6)
    elif args.exchange == "ghost":
        # square workloads only: x partitioned like the rows, x <- alpha * A * x, each rank receiving just the entries its
        # columns reference (BASELINE configs[4]: 4 + 3 doubles per neighbour instead of an allgather of 256 MB slices)
        from spmv_acc_amd.dist import GhostedRowShardedSpmv

        if n != world * m:
            raise SystemExit("--exchange ghost needs a square workload (--workload banded)")
        bounds = np.arange(world + 1, dtype=np.int64) * m
        eng = GhostedRowShardedSpmv(rank, world, bounds, W["rp"], W["ci"], W["v"], device, strategy=strat)
        eng.set_x(x[rank * m: (rank + 1) * m])
        alpha, beta = 0.4, 0.0  # spectral radius of 0.4 * A is below 1: the iteration neither overflows nor underflows
        for _ in range(max(args.warmup, 1)):
            eng.iterate(alpha)

        def ghost_region():
            for _ in range(args.steps):
                eng.iterate(alpha)

        wall, walls = median_region(ghost_region, REGION_REPS, sync_all)
        out_extra["region_reps"] = REGION_REPS
        xe = eng.x_ext.clone()
        ms = spmv_acc_amd.time_spmv(strat, min(args.steps, 50), alpha, beta, m, eng.n_local + eng.n_ghost, nnz, W["rp"],
                                    eng.cols_local, W["v"], xe, y)
        ev_ms = float(np.mean(ms))
        t = torch.tensor([wall], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall = float(t.item())
        out_extra["exchange"] = "ghost"
        out_extra["spmv_only_gflops_per_gpu"] = round(2.0 * nnz / (ev_ms * 1e-3) / 1e9, 3)
        out_extra["exchanged_bytes_per_rank_per_step"] = eng.exchanged_bytes_per_step
        out_extra["ghost_columns"] = eng.n_ghost
    else:
        budget_s = float(os.environ.get("SPMV_ACC_BENCH_BUDGET_S", "420"))

        def time_left():
            t = torch.tensor([budget_left(budget_s)], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return float(t.item())

        wall, ev_ms, extra = sharded_leg(torch, dist, args, W, x, y0, alpha, beta, rank, world, device, backend, force_dist,
                                         args.steps, args.warmup, time_left=time_left)
        out_extra.update(extra)
        if backend == "nccl":
            out_extra["rccl"] = rccl_debug_summary(rank)
        if args.workload == "hardesty3" and args.scale == 1.0 and not args.no_legs:
            # Side legs under the run's wall-clock allowance (SPMV_ACC_BENCH_BUDGET_S, default 420 s; the decision is collective -- the minimum over
            # ranks): a slow exchange must not cost the run its line by eating the driver's timeout.
            if time_left() > 150.0:
                # BASELINE configs[4] as a first-class leg of every N > 1 run: each rank owns 32 M rows of the (N * 32 M)-row banded
                # matrix (global column ids, x replicated), beta = 0, one exchange of the y slices per step.
                from spmv_acc_amd import synth as _synth

                rows = 32_000_000
                brp, bci, bv = _synth.banded_torch(rows, first_row=rank * rows, total_rows=world * rows, device=device)
                BW = dict(m=rows, n=world * rows, nnz=int(brp[-1].item()), rp=brp, ci=bci, v=bv, strategy="adaptive")
                gen_b = torch.Generator(device=device)
                gen_b.manual_seed(4321)
                bx = torch.rand(world * rows, generator=gen_b, device=device, dtype=torch.float64) * 2 - 1
                by0 = torch.zeros(rows, dtype=torch.float64, device=device)
                bsteps = min(args.steps, 50)
                bwall, bev, bextra = sharded_leg(torch, dist, args, BW, bx, by0, 1.0, 0.0, rank, world, device, backend, force_dist,
                                                 bsteps, min(args.warmup, 5), time_left=time_left)
                b_alg_b = _synth.algorithmic_bytes(rows, rows + 7, BW["nnz"], beta_nonzero=False)  # x: the columns the shard references
                bextra.update({
                    "workload": f"banded offsets -4..+3, {world} x 32 M rows (BASELINE configs[4]; 8 ranks = the 256 M-row matrix), "
                                "row-range shards, x replicated, beta = 0, allgather(y) per step",
                    "rows_per_gpu": rows, "nnz_per_gpu": BW["nnz"], "steps": bsteps,
                    "spmv_only_frac_of_hbm_peak": round(b_alg_b / (bev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)})
                # beside it, what this matrix allows when x is partitioned like the rows (x_{k+1} = f(y_k) solvers): each rank receives only
                # the x entries its columns reference -- 4 + 3 doubles per neighbour for this band -- instead of every peer's 256 MB slice
                # (GhostedRowShardedSpmv, `--exchange ghost`).  Reported, never the headline: north_star prescribes the allgather of y.
                try:
                    if time_left() < 60.0:
                        raise RuntimeError("skipped: time budget (SPMV_ACC_BENCH_BUDGET_S)")
                    from spmv_acc_amd.dist import GhostedRowShardedSpmv

                    gbounds = np.arange(world + 1, dtype=np.int64) * rows
                    geng = GhostedRowShardedSpmv(rank, world, gbounds, brp, bci, bv, device, strategy="adaptive")
                    geng.set_x(bx[rank * rows: (rank + 1) * rows])
                    for _ in range(3):
                        geng.iterate(0.4)  # (spectral radius of 0.4 * A is below 1)
                    dist.barrier()
                    torch.cuda.synchronize()
                    g0 = time.perf_counter()
                    for _ in range(bsteps):
                        geng.iterate(0.4)
                    dist.barrier()
                    torch.cuda.synchronize()
                    gt = torch.tensor([time.perf_counter() - g0], dtype=torch.float64, device=device)
                    dist.all_reduce(gt, op=dist.ReduceOp.MAX)
                    gwall = float(gt.item())
                    bextra["halo_exchange"] = {
                        "what": "x partitioned like the rows, x <- 0.4 * A * x, each rank receiving only the columns it references",
                        "ms_per_step": round(gwall / bsteps * 1e3, 6), "gflops_total": round(2.0 * BW["nnz"] * world * bsteps / gwall / 1e9, 3),
                        "exchanged_bytes_per_rank_per_step": geng.exchanged_bytes_per_step, "ghost_columns": geng.n_ghost}
                    spmv_acc_amd.release_plans(brp)
                    del geng
                except Exception as ex:  # noqa: BLE001 -- a side leg must not cost the run its line
                    bextra["halo_exchange_error"] = repr(ex)[:200]
                out_extra["banded"] = bextra
                spmv_acc_amd.release_plans(brp)
                del brp, bci, bv, bx, by0, BW
                torch.cuda.empty_cache()
            else:
                out_extra["banded"] = {"skipped": "time budget (SPMV_ACC_BENCH_BUDGET_S)"}
            if time_left() > 90.0:
                # STRONG scaling beside the weak-scaling value ("strong scaling on a problem that fits one GPU and weak
                # scaling"): the ONE Hardesty3-sized matrix (same seed on every rank) cut into `world` nnz-balanced row ranges, equal
                # padded shards, x replicated, allgather(y) per step.  gflops_total counts the matrix once.
                from spmv_acc_amd.dist import local_csr_slice, shard_bounds

                gm, gn, gnnz, grp, gci, gv = synth.hardesty3_like_torch(device=device, seed=0xC2, scale=args.scale)
                h_rp = grp.cpu().numpy()
                sb = shard_bounds(gm, world, mode=1, h_rowptr=h_rp)
                r0, r1 = int(sb[rank]), int(sb[rank + 1])
                lrp, lci, lv = local_csr_slice(grp, gci, gv, r0, r1)
                lrp = lrp.contiguous()
                SW = dict(m=r1 - r0, n=gn, nnz=int(h_rp[r1] - h_rp[r0]), rp=lrp, ci=lci, v=lv, strategy=strat)
                gen_s = torch.Generator(device=device)
                gen_s.manual_seed(99)
                sy0 = torch.rand(r1 - r0, generator=gen_s, device=device, dtype=torch.float64)
                ssteps = min(args.steps, 100)
                swall, sev, sextra = sharded_leg(torch, dist, args, SW, x, sy0, alpha, beta, rank, world, device, backend, force_dist, ssteps,
                                                 min(args.warmup, 5), bounds=sb, time_left=time_left)
                sextra.pop("spmv_plus_exchange_gflops_total", None)  # (that key assumes equal non-zeros per rank: weak scaling)
                sextra.update({"workload": "the ONE Hardesty3-sized matrix in `world` nnz-balanced row ranges (strong scaling)",
                               "rows_this_rank": r1 - r0, "nnz_this_rank": SW["nnz"], "steps": ssteps,
                               "gflops_total": round(2.0 * gnnz * ssteps / swall / 1e9, 3),
                               "ms_per_step": round(swall / ssteps * 1e3, 6)})
                out_extra["strong_scaling"] = sextra
                spmv_acc_amd.release_plans(lrp)
                del grp, gci, gv, lrp, lci, lv, SW
                torch.cuda.empty_cache()
            else:
                out_extra["strong_scaling"] = {"skipped": "time budget (SPMV_ACC_BENCH_BUDGET_S)"}

    ms_per_step = wall / args.steps * 1e3
    nnz_total = nnz * world  # weak scaling: every rank processes its own nnz
    gflops = 2.0 * nnz_total * args.steps / wall / 1e9
    b_alg = synth.algorithmic_bytes(m, n, nnz, beta_nonzero=beta != 0.0)
    if dist_leg or args.exchange == "ghost":
        b2b_ms = ev_ms  # (N > 1: ev_ms is the per-launch event mean of the local SpMV, y not reset)
    kernel_ms = out_extra.get("kernel_clock_ms_median", ev_ms)  # (N > 1: no kernel clock, the event pair around the local SpMV)
    # roofline.achieved: the kernel's average launch duration over the TIMED REGION (the K back-to-back launches `value` is measured on, one event
    # pair around them, median repetition) -- the contract's definition, and the figure the rocprofv3 summary of the same command averages to
    # (the region's launches are most of the kernel's dispatches).  The reference harness's protocol stands beside it (per_launch_protocol,
    # kernel_clock_reset_protocol): under it the same kernel is 1-3 % slower (the y reset passes through the caches between two SpMVs).
    roofline = roofline_block(b_alg, b2b_ms, kernel_ms, ev_ms, out_extra.get("per_launch_cold_ms_median"))
    roofline.update({
        "unit_note": "algorithmic bytes / launch time; a working set of <= 256 MB is served by the Infinity Cache between "
                     "launches, so small matrices can show more than the HBM rate -- it is a rate of useful bytes, not a PMC reading",
        "protocol": "hipEvents around the timed region's K back-to-back launches on the library stream, per launch, median of the repetitions"
                    if not dist_leg else "per-launch events around the local SpMV, y not reset",
        "traffic": pmc_traffic(args.workload, strat) if args.scale == 1.0 else None,
        "traffic_lower_bound": pmc_traffic(args.workload, strat, "lower_bound_bytes") if args.scale == 1.0 else None,
        "traffic_source": "profiles/pmc_traffic.json (builder-run rocprofv3 --pmc passes, tools/profile_round.sh; not measured by this run)"})
    result = {
        "metric": "CSR SpMV GFLOP/s (fp64, int32 indices; achieved HBM GB/s in roofline)",
        "value": round(gflops, 3), "unit": "GFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 6), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": W["name"], "rows_per_gpu": m, "cols": n, "nnz_per_gpu": nnz, "strategy": strat,
                   "alpha": alpha, "beta": beta, "scale": args.scale,
                   "parallelism": "single GPU" if world == 1 else f"row-range shard x{world} + allgather(y) over {'RCCL' if backend == 'nccl' else backend}"},
        "roofline": roofline,
        "protocol": "value / ms_per_step / roofline: K back-to-back SpMVs on one matrix resident in HBM (a solver loop: y is iterated in place, never reset; "
                    "consecutive SpMVs on a plan walk the matrix in alternating directions -- library default, tunable zigzag), wall clock and event time of "
                    "the same region, median of the repetitions.  Every extra leg's us / frac and roofline.per_launch_protocol: the reference harness's "
                    "protocol (csr_spmv.hpp:66-74: y reset before each launch, one event pair per launch, median), with the kernel-clock and "
                    "back-to-back figures beside it",
        "ref_formula_gibps": round(synth.reference_bytes(m, nnz) / 2**30 / (ev_ms * 1e-3), 2),
        "gflops_kernel_only_per_gpu": round(2.0 * nnz / (kernel_ms * 1e-3) / 1e9, 3),
    }
    result.update(out_extra)
    info = spmv_acc_amd.query_plan(W["rp"], m) if args.exchange != "ghost" else None
    if info:  # what the first call measured and kept for this matrix (kernel family: 0 fixed row blocks, 1 row-block-plus, 2 flat)
        result["plan"] = {k: info[k] for k in ("stream_policy", "adaptive_family", "flat_fixup", "plus_blocks", "flat_tiles", "slab_passes", "settled", "last_kernel", "col16")}
    if rank == 0:
        result["copy_ceiling_gbs"] = round(copy_ceiling_gbs(torch, device), 1)
        # the same achieved rates against what THIS box copies at (boxes of this pool read 6.25 .. 6.66 TB/s; the large sweep stand-ins
        # follow it, and with them the >= 0.70 count): informational, `roofline.frac` stays algorithmic bytes / time / 8 TB/s
        result["roofline"]["frac_of_copy_ceiling"] = round(result["roofline"]["achieved"] / result["copy_ceiling_gbs"], 4)
    if rank == 0 and world == 1 and not args.no_legs:
        # What the per-launch protocol costs before a single byte of a matrix moves: a 256-row diagonal matrix (one workgroup's
        # worth of work) under the same two protocols.  The small sweep matrices sit a few microseconds above this floor.
        tm = 256
        trp = torch.arange(tm + 1, dtype=torch.int32, device=device)
        tci = torch.arange(tm, dtype=torch.int32, device=device)
        tv = torch.ones(tm, dtype=torch.float64, device=device)
        ty0 = torch.zeros(tm, dtype=torch.float64, device=device)
        ty = ty0.clone()
        tx = torch.ones(tm, dtype=torch.float64, device=device)
        # (vector_row: a kernel family the headline matrix does not run, so that these 300 launches of a few microseconds do not sit in the rocprofv3
        # summary of the headline kernel; the floor is the protocol's, whatever the kernel)
        floor_strat = "vector_row"
        for _ in range(5):
            spmv_acc_amd.csr_spmv(1.0, 1.0, tm, tm, tm, trp, tci, tv, tx, ty, strategy=floor_strat)
        torch.cuda.synchronize()
        f_reset, f_b2b, f_min = two_protocols(torch, floor_strat, (tm, tm, tm, trp, tci, tv), tx, ty, ty0, 50, 200)
        f_nf = nofence_ms(floor_strat, (tm, tm, tm, trp, tci, tv), tx, ty, ty0, 50)
        result["launch_floor"] = {"workload": f"256-row diagonal matrix, {floor_strat} (one workgroup)",
                                  "per_launch_reset_us_median": round(f_reset * 1e3, 2), "per_launch_reset_us_min": round(f_min * 1e3, 2),
                                  "back_to_back_us_mean": round(f_b2b * 1e3, 2),
                                  "per_launch_reset_us_median_events_without_system_fence": round(f_nf * 1e3, 2)}
        spmv_acc_amd.release_plans(trp)
    if rank == 0 and world == 1 and args.workload == "hardesty3" and args.scale == 1.0 and not args.no_sensitivity:
        # Same dimensions and row lengths, none of the stand-in's 10 % uniformly random columns (SURVEY.md 8d prescribes
        # them; the real matrix is not available): shows live how much of `roofline.frac` is the gather sector cost.
        m2, n2, nnz2 = synth.LARGE_SET["Hardesty3"]
        rp2, ci2, v2 = synth.structured_csr_torch(m2, n2, nnz2, 0xC2, device=device, far_fraction=0.0)
        y2 = y0.clone()
        for _ in range(10):
            spmv_acc_amd.csr_spmv(alpha, beta, m2, n2, nnz2, rp2, ci2, v2, x, y2, strategy=strat)
        t2, t2b, _ = two_protocols(torch, strat, (m2, n2, nnz2, rp2, ci2, v2), x, y2, y0, 30, 100, alpha, beta)
        result["sensitivity"] = {"workload": "same stand-in with far_fraction 0.0 (no random columns)",
                                 "per_launch_reset_ms_median": round(t2, 6), "back_to_back_ms_mean": round(t2b, 6),
                                 "launch_ms_mean": round(t2, 6), "achieved_gbs": round(b_alg / (t2 * 1e-3) / 1e9, 2),
                                 "frac": round(b_alg / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "frac_back_to_back": round(b_alg / (t2b * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
        spmv_acc_amd.release_plans(rp2)
        del rp2, ci2, v2, y2
    if rank == 0 and world == 1 and args.workload == "hardesty3" and args.scale == 1.0 and not args.no_legs:
        result.update(extra_legs(torch, device, (m, n, nnz, W["rp"], W["ci"], W["v"]), in_process=args.legs_in_process))
        ceiling_frac = result["copy_ceiling_gbs"] / HBM_PEAK_GBS
        for s in ("flat", "adaptive"):
            result["sweep_summary"][s]["median_frac_of_copy_ceiling"] = round(result["sweep_summary"][s]["median_frac"] / ceiling_frac, 4)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(W, x, y0, args.cpu_seconds)
    elif rank == 0:
        result["cpu_baseline"] = None
    if rank == 0:
        write_full(result)
        sys.stdout.flush()
        os.write(json_fd, (compact_line(result) + "\n").encode())
    os.close(json_fd)
    if dist_leg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
