"""bench.py's launch contract for N > 1, on CPU (no GPU needed for the launch logic): `python bench.py --gpus N` without
WORLD_SIZE starts its own N ranks as child processes and relays rank 0's ONE JSON line; launched by torch.distributed.run it
is an ordinary rank; a failing rank makes the whole call fail.  SPMV_ACC_BENCH_DRYRUN=1 replaces the GPU work by a gloo
rendezvous + one all-reduce."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "SPMV_ACC_BENCH_CHILD")}
    env.update(kw)
    return env


def test_gpus_n_without_world_size_launches_its_own_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "3", "--warmup", "1"], env=_env(SPMV_ACC_BENCH_DRYRUN="1"),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # ONE JSON line on stdout, whatever the launcher and the ranks print elsewhere
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["ranks_sum"] == 3.0  # both ranks took part in the collective
    assert out["launched_by"] == "self"
    assert "launching 2 ranks" in r.stderr


def test_eight_ranks_print_one_short_line_with_every_multi_gpu_key():
    """What the driver will run once, on a node this builder never sees: `bench.py --gpus 8`.  Eight gloo ranks on CPU go through the N > 1
    machinery that needs no GPU -- self-launch, rendezvous, the repeated K-step regions with MAX over ranks per repetition, the collective
    time-budget decision -- and rank 0's line, made by compact_line() from a record with EVERY key of a real N > 1 run, stays below 6 KB."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1"], env=_env(SPMV_ACC_BENCH_DRYRUN="1"),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 6000, (len(lines), len(lines[0]))
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["ranks_sum"] == 36.0 and out["launched_by"] == "self"
    assert out["rccl"] == {"ranks": 8, "ranks_ok": True}
    # every repetition is the MAX over ranks: rank 7's fake step sleeps 8 ms
    assert out["ms_per_step"] >= 8.0, out["ms_per_step"]
    for key in ("exchange", "spmv_only_gflops_per_gpu", "spmv_only_ms_max_over_ranks", "spmv_plus_exchange_ms_per_step",
                "spmv_plus_exchange_gflops_total", "allgather_bytes_per_rank_per_step", "dependent_step_ms_by_pipeline", "pipeline_best",
                "banded", "strong_scaling", "roofline", "cpu_baseline", "config", "region_reps"):
        assert key in out, key
    assert out["banded"]["halo_exchange"]["ms_per_step"] > 0 and out["strong_scaling"]["gflops_total"] > 0
    assert out["time_left_s"] > 0  # the collective budget decision ran (minimum over ranks)


def test_launched_by_torch_distributed_run_it_is_an_ordinary_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29617", BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=_env(SPMV_ACC_BENCH_DRYRUN="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["launched_by"] == "external launcher"
    assert "launching" not in r.stderr  # no second level of ranks


def test_a_failing_rank_fails_the_call():
    """Without the dry-run switch the ranks need a GPU; on a CPU-only host they exit non-zero -- and so must the parent, with no
    JSON line.  (On a GPU box this test has nothing to show and is skipped.)"""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("a GPU is present: the ranks would run")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "1"], env=_env(), capture_output=True,
                       text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_the_parent_of_a_self_launch_never_loads_torch_or_the_library():
    """The parent must not touch the GPU (a process that has may not start other GPU programs by exec, and must not hold the
    card while its children need it): it imports neither torch nor spmv_acc_amd."""
    code = (
        "import sys, runpy, subprocess\n"
        "class P:\n"
        "    def __init__(self, *a, **k):\n"
        "        bad = [m for m in ('torch', 'spmv_acc_amd') if m in sys.modules]\n"
        "        print('LOADED', bad)\n"
        "        raise SystemExit(0)\n"
        "subprocess.Popen = P\n"
        f"sys.argv = [{BENCH!r}, '--gpus', '4']\n"
        f"runpy.run_path({BENCH!r}, run_name='__main__')\n"
    )
    r = subprocess.run([sys.executable, "-c", code], env=_env(), capture_output=True, text=True, timeout=300)
    assert "LOADED []" in r.stdout, (r.stdout, r.stderr[-1500:])


def _bench_module():
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_under_test", BENCH)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_the_stdout_line_fits_the_drivers_tail():
    """BENCH_r03.json's `parsed` was null: the line had grown to 21 KB and the driver keeps an 8,001-byte tail.  compact_line() is what
    goes to stdout now -- checked here on round 3's own full records (N = 1 with every leg; the forced-dist N > 1 shape) and on an
    N = 8 record padded with every optional key: always < 6000 bytes, always carrying the contract's keys, roofline and cpu_baseline."""
    bench = _bench_module()
    full = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_final_runs.jsonl")).readline())
    assert len(json.dumps(full)) > 15000  # (what round 3 printed)
    text = bench.compact_line(full)
    assert len(text) < 6000 and "\n" not in text
    line = json.loads(text)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert line[key] == full[key] or isinstance(full[key], dict), key
    assert line["roofline"]["frac"] == full["roofline"]["frac"] and line["roofline"]["traffic"] == full["roofline"]["traffic"]
    assert line["cpu_baseline"]["cores"] == full["cpu_baseline"]["cores"] and line["cpu_baseline"]["value"] == full["cpu_baseline"]["value"]
    assert line["sweep"]["Hardesty3"] == [full["sweep"]["Hardesty3"]["flat"]["frac"], full["sweep"]["Hardesty3"]["adaptive"]["frac"]]
    assert line["rmat25"]["frac"] == full["rmat25"]["line_enhance"]["frac"] and line["banded_shard"]["frac"] == full["banded_shard"]["adaptive"]["frac"]

    dist = json.load(open(os.path.join(ROOT, "profiles", "r03_force_dist_one_rank_rccl.json")))
    dist["n_gpus"] = 8
    leg = {k: dist[k] for k in ("exchange", "spmv_only_gflops_per_gpu", "spmv_only_ms_max_over_ranks", "spmv_plus_exchange_ms_per_step",
                                "spmv_plus_exchange_gflops_total", "allgather_bytes_per_rank_per_step", "dependent_step_ms_by_pipeline")}
    dist["banded"] = dict(leg, workload="x" * 300, rows_per_gpu=32_000_000, nnz_per_gpu=256_000_000, steps=50, spmv_only_frac_of_hbm_peak=0.7,
                          exchange_ms={"allgather": 1.0, "p2p": 1.1}, halo_exchange={"what": "y" * 200, "ms_per_step": 0.7, "gflops_total": 5000.0})
    dist["strong_scaling"] = dict(leg, workload="z" * 200, gflops_total=3000.0, ms_per_step=0.03)
    dist["rccl"]["lines"] = ["NCCL INFO " + "w" * 150] * 8
    text = bench.compact_line(dist)
    assert len(text) < 6000
    line = json.loads(text)
    assert line["rccl"]["ranks"] == dist["rccl"]["nranks_seen"] and line["spmv_plus_exchange_ms_per_step"] == dist["spmv_plus_exchange_ms_per_step"]
    assert line["banded"]["spmv_only_frac_of_hbm_peak"] == 0.7 and line["strong_scaling"]["gflops_total"] == 3000.0
    assert "roofline" in line and "cpu_baseline" in line


def test_roofline_frac_keeps_its_source():
    """the protocol stopped being a lever in round 5 and stays frozen -- `roofline.frac` / `achieved` are the timed region's back-to-back
    figure, every extra leg's `us` / `frac` and every `ge_0.70` count are the reference harness's per-launch protocol (benchmark/csr_spmv.hpp:66-74,
    benchmark/utils/benchmark_time.cpp:23-43), the kernel clock and the cold-cache column stand beside them.  This test fails when the definition table,
    the field a figure is computed from, or the keys of the line change: any further movement of a gate has to come from a kernel's microseconds."""
    bench = _bench_module()
    assert bench.ROOFLINE_DEFINITION == {
        "version": "r05", "frac": "back_to_back",
        "achieved": "algorithmic bytes (12 nnz + 4 (m + 1) + 8 n + 16 m; 8 m for y at beta = 0) / back_to_back launch time",
        "legs_and_gates": "per_launch",
        "earlier_rounds": "rounds 1-4 quoted roofline.frac on per_launch: compare their lines with roofline.per_launch_protocol.frac (r04's 0.5461 -> this key)"}
    assert bench.HBM_PEAK_GBS == 8000.0 and bench.REGION_REPS == 7
    b_alg = 710_508_500  # the headline stand-in's algorithmic bytes (12 * 40,451,632 + 4 * 8,217,821 + 8 * 7,591,564 + 16 * 8,217,820)
    r = bench.roofline_block(b_alg, b2b_ms=0.150, kernel_ms=0.155, ev_ms=0.160, cold=0.170)
    assert r["frac"] == round(b_alg / 0.150e-3 / 1e9 / 8000.0, 4) == r["back_to_back"]["frac"] and r["launch_ms_mean"] == 0.150
    assert abs(r["achieved"] - b_alg / 0.150e-3 / 1e9) < 0.01
    assert r["per_launch_protocol"]["launch_ms_median"] == 0.160 and r["per_launch_protocol"]["frac"] == round(b_alg / 0.160e-3 / 1e9 / 8000.0, 4)
    assert r["kernel_clock_reset_protocol"]["launch_ms_median"] == 0.155 and r["kernel_clock_reset_protocol"]["frac"] == round(b_alg / 0.155e-3 / 1e9 / 8000.0, 4)
    assert r["cold_protocol"]["launch_ms_median"] == 0.170 and r["frac_cold"] == round(b_alg / 0.170e-3 / 1e9 / 8000.0, 4) < r["per_launch_protocol"]["frac"]
    assert r["definition"]["frac"] == "back_to_back" and abs(r["cached_share_of_frac"] - (1 - 0.160 / 0.170)) < 2e-3
    # no other input moves frac: the three other clocks may be anything
    assert bench.roofline_block(b_alg, 0.150, 9.0, 9.0, 9.0)["frac"] == r["frac"] and bench.roofline_block(b_alg, 0.150, 9.0, 9.0)["frac"] == r["frac"]
    assert "frac_cold" not in bench.roofline_block(b_alg, 0.150, 0.155, 0.160)
    # the line carries the definition's short form, the cold column and the cold count next to the gate's count
    full = json.loads(open(os.path.join(ROOT, "profiles", "r03_bench_final_runs.jsonl")).readline())
    full["roofline"] = dict(r, traffic=None, traffic_lower_bound=None)
    for s in ("flat", "adaptive"):
        full["sweep_summary"][s].update({"ge_0.70_cold": 7, "median_frac_cold": 0.71, "stand_ins_on_16_bit_columns": 4})
    line = json.loads(bench.compact_line(full))
    assert line["roofline"]["definition"] == {"version": "r05", "frac": "back_to_back", "legs_and_gates": "per_launch"}
    assert line["roofline"]["frac_cold"] == r["frac_cold"] and line["sweep_summary"]["flat"]["ge_0.70_cold"] == 7
    # the legs' `frac` is computed from the per-launch median and from nothing else (timed_leg)
    src = open(BENCH).read()
    assert '"us": round(reset_ms * 1e3, 2), "frac": frac(reset_ms)' in src
    assert 'roofline = roofline_block(b_alg, b2b_ms, kernel_ms, ev_ms, out_extra.get("per_launch_cold_ms_median"))' in src
    assert '"ge_0.70": sum(1 for r in sweep.values() if r[s]["frac"] >= 0.70)' in src
