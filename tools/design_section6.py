#!/usr/bin/env python3
"""DESIGN.md section 6's numbers from bench.py's full records (bench_full.json of several runs): ranges over the runs given.
    python tools/design_section6.py profiles/r06_bench_builder_run_{5,6,7}_full.json
Prints the headline rows, the configs[2] sweep table (per launch: us, fraction; kernel clock; cold; record ints of the 16-bit column encoding the
plan kept, 0 = colindex), the summary counts, and the configs[3] / configs[4] lines -- to be pasted, not parsed."""
import json
import sys

runs = [json.load(open(p)) for p in sys.argv[1:]]


def rng(vals, nd=3):
    vals = [v for v in vals if v is not None]
    if not vals:
        return "n/a"
    lo, hi = min(vals), max(vals)
    f = f"{{:.{nd}f}}"
    return f.format(lo) if f.format(lo) == f.format(hi) else f"{f.format(lo)}–{f.format(hi)}"


def ints(vals):
    vals = sorted(set(int(v) for v in vals if v is not None))
    return "/".join(str(v) for v in vals)


print(f"runs: {len(runs)}")
print("value", rng([r["value"] for r in runs], 1), "GFLOP/s; ms_per_step", rng([r["ms_per_step"] for r in runs], 5))
ro = [r["roofline"] for r in runs]
print("roofline.frac (back to back)", rng([x["frac"] for x in ro]), "launch_ms_mean", rng([x["launch_ms_mean"] for x in ro], 4))
print("per launch", rng([x["per_launch_protocol"]["frac"] for x in ro]), "kernel clock", rng([x["kernel_clock_reset_protocol"]["frac"] for x in ro]),
      "cold", rng([x.get("frac_cold") for x in ro]), "cached share", rng([x.get("cached_share_of_frac") for x in ro]))
print("traffic", sorted(set(x.get("traffic") for x in ro)), "algorithmic", sorted(set(x.get("algorithmic_bytes_per_launch") for x in ro)))
print("first call ms", rng([r.get("first_call_ms") for r in runs], 1), "settle rest ms", rng([r.get("settle_rest_ms") for r in runs], 1))
cb = [r.get("cpu_baseline") or {} for r in runs]
print("cpu baseline", rng([c.get("value") for c in cb], 1), "GFLOP/s on", ints([c.get("cores") for c in cb]), "threads")
print()
print("| stand-in | flat | adaptive | adaptive on colindex only (`col16 = 0`) | flat's tile kernel alone |")
print("|---|---|---|---|---|")
for name in runs[0]["sweep"]:
    cells = []
    for s in ("flat", "adaptive"):
        legs = [r["sweep"][name][s] for r in runs]
        cells.append(f"{rng([l['us'] for l in legs], 1)} ({rng([l['frac'] for l in legs])}; kernel clock {rng([l.get('frac_kernel_clock') for l in legs])}; "
                     f"cold {rng([l.get('frac_cold') for l in legs])}; col16 {ints([l.get('col16') for l in legs])})")
    for s in ("adaptive_colindex_only", "flat_tile_kernel"):
        cells.append(rng([r["sweep"][name].get(s, {}).get("frac") for r in runs]))
    print(f"| {name} | " + " | ".join(cells) + " |")
print()
for s in ("flat", "adaptive", "flat_tile_kernel", "adaptive_colindex_only"):
    ss = [r["sweep_summary"].get(s) for r in runs if r["sweep_summary"].get(s)]
    if not ss:
        continue
    keys = ("ge_0.70", "ge_0.70_kernel_clock", "ge_0.70_back_to_back", "ge_0.70_cold", "stand_ins_on_16_bit_columns")
    print(s, {k: ints([x.get(k) for x in ss]) for k in keys},
          {k: rng([x.get(k) for x in ss]) for k in ("median_frac", "median_frac_kernel_clock", "median_frac_back_to_back", "median_frac_cold", "median_cached_share_of_frac", "min_frac")})
print()
for leg in ("rmat25", "banded_shard"):
    for s in sorted(set(k for r in runs for k, v in r.get(leg, {}).items() if isinstance(v, dict) and "us" in v)):
        legs = [r[leg][s] for r in runs if s in r.get(leg, {})]
        print(leg, s, "us", rng([l["us"] for l in legs], 1), "frac", rng([l["frac"] for l in legs]), "kernel clock", rng([l.get("frac_kernel_clock") for l in legs]),
              "cold", rng([l.get("frac_cold") for l in legs]), "kernel", sorted(set(str(l.get("kernel")) for l in legs)))
