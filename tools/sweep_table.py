#!/usr/bin/env python3
"""Markdown table of bench.py's `sweep` leg (BASELINE configs[2]) from one or more bench JSON lines:
    python tools/sweep_table.py gpurun_out/r2_bench_b.json gpurun_out/r2_bench_c.json ... > profiles/r02_large_set_sweep.md"""
import json, sys
runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in sys.argv[1:]]
names = list(runs[0]["sweep"])
print("# BASELINE configs[2] sweep on MI355X, measured by bench.py's `sweep` leg (live, same process as the headline line).")
print("# One column group per run (each run = a fresh MI355X box: the boxes differ by up to ~8 % on the 150 us kernels).")
print("# us = mean of 60-200 back-to-back launches between one hipEvent pair; frac = algorithmic bytes (12 nnz + 4 (m+1) + 8 n + 16 m) / us / 8 TB/s.")
print("# flat = the strategy BASELINE names; adaptive = the library's timed choice; flat+col16 = opt-in 16-bit column stream (never the default).")
print()
hdr = "| matrix (stand-in) | rows | nnz |" + "".join(f" flat us (frac) r{i+1} | adaptive us (frac) r{i+1} | flat+col16 us (frac) r{i+1} |" for i in range(len(runs)))
print(hdr)
print("|" + "---|" * (3 + 3 * len(runs)))
for n in names:
    r0 = runs[0]["sweep"][n]
    line = f"| {n} | {r0['rows']} | {r0['nnz']} |"
    for d in runs:
        r = d["sweep"][n]
        for s in ("flat", "adaptive", "flat_col16_opt_in"):
            line += f" {r[s]['us']:.1f} ({r[s]['frac']:.3f}) |" if s in r else " - |"
    print(line)
print()
for i, d in enumerate(runs):
    ss = d["sweep_summary"]
    print(f"run {i+1}: flat >= 0.70 on {ss['flat']['ge_0.70']} of 12 (median {ss['flat']['median_frac']:.3f}), adaptive >= 0.70 on {ss['adaptive']['ge_0.70']} of 12 "
          f"(median {ss['adaptive']['median_frac']:.3f}); headline {d['value']} GFLOP/s, frac {d['roofline']['frac']}; copy ceiling {d['copy_ceiling_gbs']} GB/s; "
          f"R-MAT 25 line_enhance {d['rmat25']['line_enhance']['us']} us ({d['rmat25']['line_enhance']['frac']}); banded shard {d['banded_shard']['adaptive']['us']} us ({d['banded_shard']['adaptive']['frac']})")
