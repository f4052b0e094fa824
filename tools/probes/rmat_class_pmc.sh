#!/bin/bash
# Runs ON THE GPU BOX: L2 hit / miss counters per kernel for one row class of R-MAT 25 under S = 8 passes.   usage: rmat_class_pmc.sh <class lower bound> <tag> "<counters>"
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_class_$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLASS_ONLY=$1
timeout -k 10 500 rocprofv3 --pmc $3 --output-format csv -d $OUT/pmc -- python3 $R/tools/probes/rmat_class_probe.py 25 > $OUT/run.txt 2> $OUT/pmc.log || exit 1
F=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    if "segment_long" in row["Kernel_Name"] or "segment_tile" in row["Kernel_Name"]:
        key = (row["Kernel_Name"].split("::")[-1].split("(")[0], row["Grid_Size"] if "Grid_Size" in row else row.get("Grid_Size_X", "?"))
        acc[key][row["Counter_Name"]].append(float(row["Counter_Value"]))
for key, ctr in sorted(acc.items(), key=lambda kv: -max(len(v) for v in kv[1].values())):
    print(key, {c: round(sum(v) / len(v), 1) for c, v in ctr.items()}, "dispatches", max(len(v) for v in ctr.values()))
PY
rm -rf $OUT/pmc
