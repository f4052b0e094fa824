#!/usr/bin/env python3
"""The measurement tables of DESIGN.md section 6 from bench.py JSON lines (ranges over the runs given):
    python tools/design_tables.py gpurun_out/r3_bench7.json gpurun_out/r3_bench8.json ...      (one line per file, or)
    python tools/design_tables.py profiles/r03_bench_final_runs.jsonl                          (one run per line)"""
import json, sys
runs = [json.loads(line) for p in sys.argv[1:] for line in open(p) if line.strip().startswith("{")]
def rng(vals, nd=3):
    vals = [v for v in vals if v is not None]
    lo, hi = min(vals), max(vals)
    f = f"{{:.{nd}f}}"
    return f.format(lo) if f.format(lo) == f.format(hi) else f"{f.format(lo)}–{f.format(hi)}"
print(f"runs: {len(runs)}")
print("headline value", rng([r["value"] for r in runs], 0), "ms/step", rng([r["ms_per_step"] for r in runs], 4))
print("reset ms", rng([r["roofline"]["launch_ms_mean"] for r in runs], 4), "frac", rng([r["roofline"]["frac"] for r in runs]), "TB/s", rng([r["roofline"]["achieved"] / 1e3 for r in runs], 2))
print("b2b ms", rng([r["roofline"]["back_to_back"]["launch_ms_mean"] for r in runs], 4), "frac", rng([r["roofline"]["back_to_back"]["frac"] for r in runs]))
print("nofence ms", rng([r.get("per_launch_reset_ms_median_events_without_system_fence") for r in runs], 4))
print("far0", rng([r["sensitivity"]["frac"] for r in runs]), rng([r["sensitivity"]["frac_back_to_back"] for r in runs]))
print("floor", rng([r["launch_floor"]["per_launch_reset_us_median"] for r in runs], 2), "b2b", rng([r["launch_floor"]["back_to_back_us_mean"] for r in runs], 2), "nofence", rng([r["launch_floor"]["per_launch_reset_us_median_events_without_system_fence"] for r in runs], 2))
print("| stand-in | flat µs | flat frac (b2b) | adaptive µs | adaptive frac (b2b) | adaptive µs, fence-free events |")
print("|---|---|---|---|---|---|")
for name in runs[0]["sweep"]:
    cells = []
    for s in ("flat", "adaptive"):
        cells.append(rng([r["sweep"][name][s]["us"] for r in runs], 1))
        cells.append(f"{rng([r['sweep'][name][s]['frac'] for r in runs])} ({rng([r['sweep'][name][s]['frac_back_to_back'] for r in runs])})")
    cells.append(f"{rng([r['sweep'][name]['adaptive']['us_events_without_system_fence'] for r in runs], 1)} ({rng([r['sweep'][name]['adaptive']['frac_events_without_system_fence'] for r in runs])})")
    print(f"| {name} | " + " | ".join(cells) + " |")
for s in ("flat", "adaptive"):
    print(s, "ge_0.70", [r["sweep_summary"][s]["ge_0.70"] for r in runs], "b2b", [r["sweep_summary"][s]["ge_0.70_back_to_back"] for r in runs],
          "median", rng([r["sweep_summary"][s]["median_frac"] for r in runs]), "b2b median", rng([r["sweep_summary"][s]["median_frac_back_to_back"] for r in runs]))
print("rmat ms", rng([r["rmat25"]["line_enhance"]["us"] / 1e3 for r in runs], 2), "frac", rng([r["rmat25"]["line_enhance"]["frac"] for r in runs]))
if all("line_enhance_without_slab_passes" in r["rmat25"] for r in runs):
    print("rmat one-kernel ms", rng([r["rmat25"]["line_enhance_without_slab_passes"]["us"] / 1e3 for r in runs], 2), "frac", rng([r["rmat25"]["line_enhance_without_slab_passes"]["frac"] for r in runs]), "path", {r["rmat25"].get("path", "")[:40] for r in runs})
print("rmat slabs ms", rng([r["rmat25"]["line_enhance_col_slabs8_opt_in"]["us"] / 1e3 for r in runs], 2), "frac", rng([r["rmat25"]["line_enhance_col_slabs8_opt_in"]["frac"] for r in runs]))
print("banded us", rng([r["banded_shard"]["adaptive"]["us"] for r in runs], 0), "frac", rng([r["banded_shard"]["adaptive"]["frac"] for r in runs]), "b2b", rng([r["banded_shard"]["adaptive"]["frac_back_to_back"] for r in runs]))
print("cpu", [(r.get("cpu_baseline") or {}).get("value") for r in runs], [(r.get("cpu_baseline") or {}).get("value_1thread") for r in runs])
print("copy ceiling", [r["copy_ceiling_gbs"] for r in runs])
