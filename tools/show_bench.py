#!/usr/bin/env python3
"""Print the table of one bench.py JSON line (headline, launch floor, sweep under both protocols, R-MAT, banded shard)."""
import json, sys
d = json.load(open(sys.argv[1]))
r = d["roofline"]
print(f"headline: value {d['value']} GFLOP/s  ms/step {d['ms_per_step']}  reset {r['launch_ms_mean']} ms frac {r['frac']}  b2b {r['back_to_back']['launch_ms_mean']} frac {r['back_to_back']['frac']}  plan {d.get('plan')}")
if "launch_floor" in d: print("launch floor:", d["launch_floor"])
if "sensitivity" in d: print("far=0:", d["sensitivity"]["frac"], d["sensitivity"].get("frac_back_to_back"))
if "sweep" in d:
    for k, v in d["sweep"].items():
        print(f"{k:18s} flat {v['flat']['us']:8.2f} {v['flat']['frac']:.3f} ({v['flat']['frac_back_to_back']:.3f}) {v['flat']['plan']} | adaptive {v['adaptive']['us']:8.2f} {v['adaptive']['frac']:.3f} ({v['adaptive']['frac_back_to_back']:.3f}) {v['adaptive']['plan']}")
    for s in ("flat", "adaptive"):
        ss = d["sweep_summary"][s]
        print(s, "ge_0.70", ss["ge_0.70"], "median", round(ss["median_frac"], 4), "| b2b ge_0.70", ss["ge_0.70_back_to_back"], "median", round(ss["median_frac_back_to_back"], 4))
    print("rmat25", d["rmat25"]["line_enhance"]["us"], d["rmat25"]["line_enhance"]["frac"])
    print("banded", d["banded_shard"]["adaptive"]["us"], d["banded_shard"]["adaptive"]["frac"])
if d.get("cpu_baseline"): print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"]["value_1thread"])
