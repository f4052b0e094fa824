#!/bin/bash
# Runs ON THE GPU BOX: per-kernel times of one row class of R-MAT 25 under S = 8 passes (profiles/probes/rmat_class_probe.py under rocprofv3).
# usage: profiles/probes/rmat_class_profile.sh <class lower bound> <tag>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profile_class_$2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export CLASS_ONLY=$1
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $R/profiles/probes/rmat_class_probe.py 25 > $OUT/run.txt 2> $OUT/trace.log || exit 1
T=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python3 - $T <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "segment_" in r["Kernel_Name"] or "scale_y" in r["Kernel_Name"]]
# the last SpMV's dispatches, in order
tail = rows[-40:]
for r in tail:
    n = r["Kernel_Name"].replace("spmv_acc::(anonymous namespace)::", "").split("(")[0]
    print(f"{n:28s} grid {int(r['Grid_Size_X'])//256:8d} wgs  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:9.1f} us")
PY
rm -rf $OUT/trace
grep -v amdgpu $OUT/run.txt | cut -c1-200
