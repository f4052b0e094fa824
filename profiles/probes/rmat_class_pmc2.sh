# LDS / issue counters of the slab passes' tile kernel on R-MAT 25 (rocprofv3 --pmc, one group per pass of the program)
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in "SQ_LDS_BANK_CONFLICT SQ_LDS_ACTIVE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_INSTS_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_VMEM_RD SQ_INSTS_SALU"; do
  OUT=$R/gpurun_out/pmc_seg/$(echo $C | tr ' ' '_')
  mkdir -p $OUT
  timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $OUT -- python3 $R/profiles/probes/rmat_time.py > $OUT/run.log 2>&1 || { echo "fail $C"; tail -3 $OUT/run.log; continue; }
  F=$(find $OUT -name "*counter_collection.csv" | head -1)
  python3 - "$F" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "segment_tile" in k:
        acc[("hinted" if "<true>" in k or "Lb1" in k else "plain", row["Counter_Name"])].append(float(row["Counter_Value"]))
for (k, c), v in sorted(acc.items()):
    print(k, c, "n=%d" % len(v), "mean=%.4g" % (sum(v) / len(v)), "max=%.4g" % max(v))
PY
  rm -rf $OUT/*/ 2>/dev/null
done
