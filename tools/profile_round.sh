#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 passes over bench.py, summaries into gpurun_out/profile_<tag>/.
#   1. --kernel-trace --stats   (per-kernel durations of the same command bench.py times)
#   2. --pmc FETCH_SIZE         (own pass)
#   3. --pmc WRITE_SIZE         (own pass; FETCH_SIZE needs 3 of the 4 TCC slots)
# usage: tools/profile_round.sh <tag> [bench.py args...]
set -o pipefail
TAG=${1:-r01}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py "$@" --no-cpu-baseline > $OUT/bench_under_trace.json 2> $OUT/trace.log || exit 1
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
{ head -1 $S; grep -E "spmv_acc" $S; } > $OUT/kernel_stats_spmv.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py "$@" --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_under_$c.json 2> $OUT/pmc_$c.log || exit 1
  F=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$c" > $OUT/pmc_$c.summary.txt <<'PY'
import csv, sys, collections
f, c = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for row in csv.DictReader(open(f)):
    if "spmv_acc" in row["Kernel_Name"] and row["Counter_Name"] == c:
        name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        acc[name.split("(")[0]].append(float(row["Counter_Value"]))
for k, v in acc.items():
    print(f"{c} kernel={k} dispatches={len(v)} mean_KB={sum(v)/len(v):.1f} min_KB={min(v):.1f} max_KB={max(v):.1f}")
PY
done
rm -rf $OUT/trace/*/*_kernel_trace.csv $OUT/pmc_*/  # keep summaries only (small)
cat $OUT/kernel_stats_spmv.csv $OUT/pmc_*.summary.txt
