#!/bin/bash
# HBM-side traffic of one strategy on several workloads: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (gfx950: FETCH_SIZE
# tallies wide coalesced reads at half their bytes -- MI355X_MICROARCH.md; corrected bytes = (2*FETCH + WRITE) * 1024).
# usage (on the GPU box): tools/pmc_traffic.sh <strategy> <workload>...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
S=${1:-adaptive}; shift
cd /tmp && export TMPDIR=/tmp
for W in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    OUT=$R/gpurun_out/pmc_traffic/$W/$C
    mkdir -p $OUT
    timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 $R/tools/pmc_run.py --workload $W --strategy $S --iters 12 > $OUT/run.log 2>&1 || { echo "fail $W $C"; tail -3 $OUT/run.log; continue; }
    F=$(find $OUT -name "*counter_collection.csv" | head -1)
    python3 - "$F" "$W" "$C" "$OUT/run.log" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if ("rowblock_stream" in k or "flat_tile" in k or "plus_kernel" in k) and row["Counter_Name"] == sys.argv[3]:
        acc[k.split("(")[0].replace("void spmv_acc::(anonymous namespace)::", "")].append(float(row["Counter_Value"]))
balg = open(sys.argv[4]).read().split()[-1]
for k, v in sorted(acc.items(), key=lambda kv: -len(kv[1]))[:1]:  # the variant the plan settled on (most dispatches)
    tail = v[-4:]
    print(sys.argv[2], k, sys.argv[3], "dispatches=%d" % len(v), "KB(last 4 mean)=%.1f" % (sum(tail) / len(tail)), "B_alg(beta=1)=%s" % balg)
PY
    rm -rf $OUT/*/ 2>/dev/null
  done
done
