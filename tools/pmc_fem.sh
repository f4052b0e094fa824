#!/bin/bash
# Counter-level evidence for the FEM-class kernels of the configs[2] sweep: who owns the gap between the tile kernels
# (~5.7 TB/s of algorithmic bytes) and a bare read loop (~7.1 TB/s)?  Runs ON THE GPU BOX (via gpurun):
#   0. an unprofiled run: the plan the library settles on + per-launch / back-to-back event times;
#   1. rocprofv3 --kernel-trace --stats: the kernel's own duration under the per-launch protocol (event time minus this = the protocol's floor);
#   2. one --pmc pass per counter group (never combined with a trace), the plan pinned to what pass 0 settled on.
# usage: tools/pmc_fem.sh <tag> <strategy> <workload>...     e.g.  tools/pmc_fem.sh r05 flat Bump_2911 Cube_Coup_dt6
set -o pipefail
TAG=${1:-r05}; S=${2:-flat}; shift; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
GROUPS_SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY"
GROUPS_SQ2="SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_VALU"
GROUPS_SQ3="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU"
GROUPS_SQ4="SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
GROUPS_TC1="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
GROUPS_TC2="TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_READ_sum"
GROUPS_TC3="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum"
GROUPS_TC4="TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum"
GROUPS_TC5="TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_READ_SECTORS_sum"
for W in "$@"; do
  OUT=$R/gpurun_out/pmc_fem_$TAG/${W}_$S
  mkdir -p $OUT
  # EXTRA_TUNABLES (e.g. strict_strategy=1): set for every pass, the unprofiled one included
  export SPMV_ACC_TUNABLES="${EXTRA_TUNABLES:-}"
  python3 $R/tools/pmc_fem_run.py --workload $W --strategy $S --iters 40 2> $OUT/plain.log | tail -1 > $OUT/plain.json || { echo "fail plain $W"; tail -3 $OUT/plain.log; continue; }
  cat $OUT/plain.json
  PIN=$(python3 - "$OUT/plain.json" "$S" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["plan"]
pins = ["stream_plain=%d" % d["stream_policy"]] if d.get("stream_policy", -1) >= 0 else []
print(",".join(pins))
PY
)
  export SPMV_ACC_TUNABLES="$PIN${EXTRA_TUNABLES:+,$EXTRA_TUNABLES}"
  echo "[pmc_fem] $W $S pinned: $SPMV_ACC_TUNABLES"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/tools/pmc_fem_run.py --workload $W --strategy $S --iters 40 > $OUT/trace.json 2> $OUT/trace.log || { echo "fail trace $W"; tail -3 $OUT/trace.log; }
  ST=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
  [ -n "$ST" ] && { head -1 $ST; grep -E "spmv_acc" $ST | head -6; } > $OUT/kernel_stats_spmv.csv
  rm -rf $OUT/trace
  i=0
  # PMC_LEAN=1: the traffic counters only (request sizes, FETCH_SIZE, WRITE_SIZE) -- the per-stand-in table of moved bytes over all twelve
  # PMC_LEAN=2: the wait split + the traffic counters (round 6: the 16-bit column kernel); PMC_LEAN=3: no counter pass (timing + kernel trace only)
  if [ "${PMC_LEAN:-0}" = "1" ]; then PASSES=("$GROUPS_TC4" "FETCH_SIZE" "WRITE_SIZE");
  elif [ "${PMC_LEAN:-0}" = "2" ]; then PASSES=("$GROUPS_SQ1" "$GROUPS_SQ2" "$GROUPS_TC4" "FETCH_SIZE" "WRITE_SIZE");
  elif [ "${PMC_LEAN:-0}" = "3" ]; then PASSES=(); else
    PASSES=("$GROUPS_SQ1" "$GROUPS_SQ2" "$GROUPS_SQ3" "$GROUPS_SQ4" "$GROUPS_TC1" "$GROUPS_TC2" "$GROUPS_TC3" "$GROUPS_TC4" "$GROUPS_TC5" "FETCH_SIZE" "WRITE_SIZE"); fi
  for C in "${PASSES[@]}"; do
    i=$((i+1))
    D=$OUT/pmc_$i
    mkdir -p $D
    timeout -k 10 300 rocprofv3 --pmc $C --output-format csv -d $D -- python3 $R/tools/pmc_fem_run.py --workload $W --strategy $S --iters 12 --no-timing > $D/run.log 2>&1 || { echo "fail group $i ($C)"; tail -2 $D/run.log; rm -rf $D; continue; }
    F=$(find $D -name "*counter_collection.csv" | head -1)
    python3 - "$F" >> $OUT/counters.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if any(s in k for s in ("rowblock_stream", "flat_tile", "flat_fixup", "plus_kernel")):
        name = k.replace("void ", "").replace("spmv_acc::(anonymous namespace)::", "").split("(")[0]
        acc[(name, row["Counter_Name"])].append(float(row["Counter_Value"]))
top = max((len(v) for v in acc.values()), default=0)
for (k, c), v in sorted(acc.items()):
    if len(v) * 2 >= top:  # the kernels of the settled plan (plan-time trials of other variants have a few dispatches each)
        tail = v[-8:]
        print(f"{k} {c} dispatches={len(v)} mean_last8={sum(tail)/len(tail):.6g}")
PY
    rm -rf $D
  done
  unset SPMV_ACC_TUNABLES
  echo "== $W $S"; cat $OUT/kernel_stats_spmv.csv; [ -f $OUT/counters.txt ] && cat $OUT/counters.txt; true
done
