#!/bin/bash
# Round 6: HBM-side traffic of ONE SpMV of R-MAT 25 under line_enhance on the default path (the slab-major copy) and on the run-list passes alone
# (col_slabs = 0).  An SpMV is many launches there, so the counter is summed over EVERY launch of the library in the process, once with 4 and once
# with 12 steady-state SpMVs after the same preparation (choices adopted from a tune cache written by an unprofiled run): (sum_12 - sum_4) / 8 is one SpMV's share, whatever its launch structure.
# Separate rocprofv3 --pmc passes for FETCH_SIZE and WRITE_SIZE, plus a --kernel-trace --stats pass.  usage (on the GPU box): tools/pmc_rmat.sh <tag>
set -o pipefail
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for MODE in copy passes; do
  OUT=$R/gpurun_out/pmc_rmat_$TAG/$MODE
  rm -rf $OUT; mkdir -p $OUT
  if [ $MODE = passes ]; then export SPMV_ACC_TUNABLES="col_slabs=0"; else unset SPMV_ACC_TUNABLES; fi
  # the timed choices once, unprofiled, into a tune cache: the profiled runs adopt them and repeat exactly the same preparation (the number of
  # trial launches of a timing phase depends on what it measures, and with it the sums below)
  export SPMV_ACC_TUNE_CACHE=$OUT/tune_cache.txt
  python3 $R/tools/pmc_fem_run.py --workload rmat25 --strategy line_enhance --iters 2 --no-timing > $OUT/warm.json 2> $OUT/warm.log || { echo "fail warm $MODE"; tail -3 $OUT/warm.log; }
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/pmc_fem_run.py --workload rmat25 --strategy line_enhance --iters 8 > $OUT/trace.json 2> $OUT/trace.log || { echo "fail trace $MODE"; tail -3 $OUT/trace.log; }
  ST=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$ST" ] && { head -1 $ST; grep -E "spmv_acc" $ST | head -10; } > $OUT/kernel_stats_spmv.csv
  rm -rf $OUT/trace
  for C in FETCH_SIZE WRITE_SIZE; do
    for N in 4 12; do
      D=$OUT/pmc_${C}_$N
      mkdir -p $D
      timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/tools/pmc_fem_run.py --workload rmat25 --strategy line_enhance --iters $N --no-timing > $D/run.log 2>&1 || { echo "fail $MODE $C $N"; tail -2 $D/run.log; continue; }
      F=$(find $D -name "*counter_collection.csv" | head -1)
      python3 - "$F" "$C" "$N" >> $OUT/sums.txt <<'PY'
import csv, sys
total = sum(float(r["Counter_Value"]) for r in csv.DictReader(open(sys.argv[1])) if r["Counter_Name"] == sys.argv[2] and "spmv_acc" in r["Kernel_Name"])
print(sys.argv[2], sys.argv[3], total)
PY
      rm -rf $D
    done
  done
  python3 - "$OUT/sums.txt" "$OUT/trace.json" "$MODE" > $OUT/per_spmv.json <<'PY'
import json, sys
s = {}
for ln in open(sys.argv[1]):
    c, n, v = ln.split()
    s[(c, int(n))] = float(v)
info = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
fetch = (s[("FETCH_SIZE", 12)] - s[("FETCH_SIZE", 4)]) / 8
write = (s[("WRITE_SIZE", 12)] - s[("WRITE_SIZE", 4)]) / 8
corrected = (2 * fetch + write) * 1024
print(json.dumps({"mode": sys.argv[3], "FETCH_SIZE_KB_per_spmv": round(fetch, 1), "WRITE_SIZE_KB_per_spmv": round(write, 1), "corrected_bytes_per_spmv": int(corrected),
                  "algorithmic_bytes": info["algorithmic_bytes"], "moved_over_algorithmic": round(corrected / info["algorithmic_bytes"], 3),
                  "kernel_clock_us_median": info["kernel_clock_us_median"], "per_launch_us_median": info["per_launch_us_median"],
                  "moved_TBps": round(corrected / info["kernel_clock_us_median"] / 1e6, 3), "last_kernel": info["plan"]["last_kernel"]}))
PY
  echo "== $MODE"; cat $OUT/per_spmv.json; cut -c1-160 $OUT/kernel_stats_spmv.csv
done
