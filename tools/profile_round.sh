#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 passes over bench.py, summaries into gpurun_out/profile_<tag>/.
#   1. --kernel-trace --stats   (per-kernel durations of the same command bench.py times)
#   2. --pmc FETCH_SIZE         (own pass)
#   3. --pmc WRITE_SIZE         (own pass; FETCH_SIZE needs 3 of the 4 TCC slots)
#   4. pmc_entry.json           (the profiles/pmc_traffic.json entry, computed from the two PMC summaries of THIS run and the
#                                algorithmic bytes bench.py printed under the trace -- tools/collect_profile.py merges it)
# usage: tools/profile_round.sh <tag> [bench.py args...]        e.g.  tools/profile_round.sh r02 --no-legs
set -o pipefail
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
echo "[profile_round] kernel trace"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $R/bench.py "$@" --no-cpu-baseline --no-sensitivity > $OUT/bench_under_trace.json 2> $OUT/trace.log || exit 1
# (bench.py measures its extra legs in child processes and rocprofv3 writes one summary per process: the parent's -- the lowest pid -- holds the timed region)
S=$(find $OUT/trace -name "*kernel_stats.csv" | awk -F/ '{f=$NF; sub(/_kernel_stats.csv/,"",f); print f+0, $0}' | sort -n | head -1 | cut -d" " -f2-)
{ head -1 $S; grep -E "spmv_acc" $S; } > $OUT/kernel_stats_spmv.csv
# The counter passes serialise and perturb the kernels, which is enough to tip the per-matrix timings between near-equal
# candidates: they run with the cache policy the (unperturbed) trace pass settled on, so all three passes profile one kernel.
POL=$(python3 -c "import json,sys; d=json.loads(open('$OUT/bench_under_trace.json').read().strip().splitlines()[-1]); print(d.get('plan',{}).get('stream_policy',-1))")
if [ "$POL" -ge 0 ] 2>/dev/null; then export SPMV_ACC_TUNABLES="stream_plain=$POL${SPMV_ACC_TUNABLES:+,$SPMV_ACC_TUNABLES}"; fi
# ... and with the column-slab passes where the trace pass's plan-time timing chose them (under the counters it can fall the other way)
SLABS=$(python3 -c "import json,sys; d=json.loads(open('$OUT/bench_under_trace.json').read().strip().splitlines()[-1]); print(d.get('plan',{}).get('slab_passes',0))")
if [ "$SLABS" -ge 2 ] 2>/dev/null; then export SPMV_ACC_TUNABLES="slab_segments=$SLABS${SPMV_ACC_TUNABLES:+,$SPMV_ACC_TUNABLES}"; fi
# ... and with the column stream the trace pass's plan settled on (round 6: the 16-bit encoding is a timed choice; col16 = its record size pins it, 0 pins colindex)
C16=$(python3 -c "import json,sys; d=json.loads(open('$OUT/bench_under_trace.json').read().strip().splitlines()[-1]); print(d.get('plan',{}).get('col16',-1))")
if [ "$C16" -ge 0 ] 2>/dev/null; then export SPMV_ACC_TUNABLES="col16=$C16${SPMV_ACC_TUNABLES:+,$SPMV_ACC_TUNABLES}"; fi
echo "[profile_round] counter passes with SPMV_ACC_TUNABLES=$SPMV_ACC_TUNABLES"
for c in FETCH_SIZE WRITE_SIZE; do
  echo "[profile_round] pmc $c"
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_$c -- python3 $R/bench.py "$@" --no-cpu-baseline --no-legs --no-sensitivity --steps 20 --warmup 5 > $OUT/bench_under_$c.json 2> $OUT/pmc_$c.log || exit 1
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
python3 - "$OUT" "$TAG" > $OUT/pmc_entry.json <<'PY'
import json, re, sys
out, tag = sys.argv[1], sys.argv[2]
def dominant(path):
    best = None
    for line in open(path):
        m = re.match(r"(\w+) kernel=(.+?) dispatches=(\d+) mean_KB=([\d.]+)", line)
        # the SpMV kernel of the timed loop: most dispatches among the tile kernels
        if m and any(k in m.group(2) for k in ("rowblock_stream", "flat_tile", "plus_kernel", "vector_tile", "vector_row", "wave_row", "direct_rows")):
            if best is None or int(m.group(3)) > best[1]:
                best = (m.group(2), int(m.group(3)), float(m.group(4)))
    return best
def slab_passes(path):
    # column-slab passes over run lists (k_segment.hip): one SpMV = S dispatches of segment_tile_kernel (+ the merge of pieces, + one
    # guard_check_kernel, whose dispatch count is therefore the number of SpMVs); per-SpMV KB = the sum over those kernels / SpMVs
    rows = {}
    for line in open(path):
        m = re.match(r"(\w+) kernel=(.+?) dispatches=(\d+) mean_KB=([\d.]+)", line)
        if m:
            rows[m.group(2)] = (int(m.group(3)), float(m.group(4)))
    tiles = [k for k in rows if "segment_tile" in k]
    checks = [k for k in rows if "guard_check" in k]
    if not tiles or not checks or rows[tiles[0]][0] < 4 * rows[checks[0]][0]:
        return None
    spmvs = rows[checks[0]][0]
    total = sum(n * kb for k, (n, kb) in rows.items() if any(s in k for s in ("segment_tile", "segment_merge", "guard_check", "scale_y")))
    return (f"spmv_acc::segment_tile_kernel x {sum(rows[k][0] for k in tiles) // spmvs} passes per SpMV (+ merge)", spmvs, total / spmvs)
f, w = dominant(f"{out}/pmc_FETCH_SIZE.summary.txt"), dominant(f"{out}/pmc_WRITE_SIZE.summary.txt")
sf, sw = slab_passes(f"{out}/pmc_FETCH_SIZE.summary.txt"), slab_passes(f"{out}/pmc_WRITE_SIZE.summary.txt")
if sf and sw:
    f, w = sf, sw
if f is None or w is None:
    sys.exit(f"profile_round: no SpMV kernel of the library found in {out}/pmc_*.summary.txt (kernel names changed? see dominant())")
bench = json.loads(open(f"{out}/bench_under_trace.json").read().strip().splitlines()[-1])
balg = bench["roofline"]["algorithmic_bytes_per_launch"]
fetch_b, write_b = f[2] * 1024.0, w[2] * 1024.0
corrected = 2.0 * fetch_b + write_b                      # MI355X_MICROARCH.md: gfx950 tallies wide coalesced reads at half
reads_alg = balg - write_b                               # algorithmic read bytes
# (until round 4 the over-fetch -- the far gathers -- was also counted at 64 B per gather as a lower bound; round 5's request-size counters show that every
# L2-missing read of these kernels, gathers included, is a 128-B request (profiles/r05_gather_request_size_microbench.txt): the corrected figure IS the traffic)
lower = corrected
print(json.dumps({"kernel": f[0], "FETCH_SIZE_KB": f[2], "WRITE_SIZE_KB": w[2], "corrected_bytes": int(round(corrected)),
                  "lower_bound_bytes": int(round(lower)), "algorithmic_bytes": balg, "round": tag,
                  "workload": bench["config"]["workload"], "strategy": bench["config"]["strategy"]}))
PY
cat $OUT/kernel_stats_spmv.csv $OUT/pmc_*.summary.txt $OUT/pmc_entry.json
