set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for W in Bump_2911 hardesty3 RM07R; do
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  OUT=$R/gpurun_out/pmc_lds/$W/$(echo $C | tr ' ' '_')
  mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 $R/tools/pmc_run.py --workload $W --iters 3 > $OUT/run.log 2>&1 || { echo "fail $W $C"; tail -3 $OUT/run.log; continue; }
  F=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$F" "$W" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "rowblock_stream" in k or "flat_tile" in k:
        acc[(k.split("(")[0][-60:], row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(sys.argv[2], k, c, "n=%d" % len(v), "last=%.4g" % v[-1])
PY
  rm -rf $OUT/*/  2>/dev/null
done; done
