#!/bin/bash
# Runs ON THE GPU BOX: FETCH_SIZE (KB, raw: double it for wide streaming reads, MI355X_MICROARCH.md) per kernel for one variant of tools/slab_segment_bench.py.
# usage: tools/hub_pmc.sh <scale> "<variant>" <tag>
set -o pipefail
SCALE=${1:-25}; VAR=${2:-hub_rows=1}; TAG=${3:-hub}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc -- python3 $R/tools/slab_segment_bench.py $SCALE line_enhance "$VAR" > $OUT/run.txt 2> $OUT/pmc.log || exit 1
F=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "spmv_acc" in row["Kernel_Name"] and row["Counter_Name"] == "FETCH_SIZE":
        name = row["Kernel_Name"].replace("spmv_acc::(anonymous namespace)::", "").replace("void ", "")
        acc[name.split("(")[0]].append(float(row["Counter_Value"]))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    if sum(v) > 1e4:
        print(f"FETCH_SIZE kernel={k[:50]:50s} dispatches={len(v):4d} sum_GB={sum(v)*1024/1e9:9.3f} mean_MB={sum(v)/len(v)*1024/1e6:10.1f} max_MB={max(v)*1024/1e6:10.1f}")
PY
rm -rf $OUT/pmc
cat $OUT/run.txt | grep -E "\{|R-MAT"
