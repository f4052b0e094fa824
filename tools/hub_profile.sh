#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of the slab passes with / without the hub windows on R-MAT (tools/slab_segment_bench.py under rocprofv3).
# usage: tools/hub_profile.sh <scale> "<variant>" <tag>
set -o pipefail
SCALE=${1:-25}; VAR=${2:-hub_rows=1}; TAG=${3:-hub}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/slab_segment_bench.py $SCALE line_enhance "$VAR" > $OUT/run.txt 2> $OUT/trace.log || exit 1
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
{ head -1 $S; grep -E "spmv_acc" $S; } > $OUT/kernel_stats_spmv.csv
rm -rf $OUT/trace/*/*_kernel_trace.csv
cat $OUT/run.txt
python3 - $OUT/kernel_stats_spmv.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].replace("spmv_acc::(anonymous namespace)::", "").split("(")[0][:60]
    print(f"{n:60s} calls {r['Calls']:>6s} total_ms {float(r['TotalDurationNs'])/1e6:10.2f} avg_us {float(r['AverageNs'])/1e3:10.1f} min_us {float(r['MinNs'])/1e3:9.1f} max_us {float(r['MaxNs'])/1e3:9.1f}")
PY
