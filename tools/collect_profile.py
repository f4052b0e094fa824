#!/usr/bin/env python3
"""Runs HERE after `gpurun -- tools/profile_round.sh <tag> ...`: copies the summaries gpurun merged back from
gpurun_out/profile_<tag>/ into profiles/ (named per round) and merges that run's pmc_entry.json into
profiles/pmc_traffic.json under "<workload key>|<strategy>", so the JSON bench.py reads and the committed PMC text come
from the same rocprofv3 passes.

    python tools/collect_profile.py r02 hardesty3
"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, key = sys.argv[1], sys.argv[2]
src = os.path.join(ROOT, "gpurun_out", f"profile_{tag}")
dst = os.path.join(ROOT, "profiles")
entry = json.load(open(os.path.join(src, "pmc_entry.json")))
strategy = entry["strategy"]
shutil.copy(os.path.join(src, "kernel_stats_spmv.csv"), os.path.join(dst, f"{tag}_{key}_{strategy}_kernel_stats.csv"))
with open(os.path.join(dst, f"{tag}_{key}_{strategy}_pmc.txt"), "w") as f:
    f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --no-legs --no-sensitivity --steps 20 --warmup 5`, "
            f"tools/profile_round.sh {tag}; per-dispatch KB\n")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f.write(open(os.path.join(src, f"pmc_{c}.summary.txt")).read())
    f.write("# entry merged into profiles/pmc_traffic.json:\n# " + json.dumps(entry) + "\n")
shutil.copy(os.path.join(src, "bench_under_trace.json"), os.path.join(dst, f"{tag}_{key}_{strategy}_bench_under_rocprof.json"))
table_path = os.path.join(dst, "pmc_traffic.json")
table = json.load(open(table_path))
table[f"{key}|{strategy}"] = entry
json.dump(table, open(table_path, "w"), indent=1)
print("merged", f"{key}|{strategy}", entry)
