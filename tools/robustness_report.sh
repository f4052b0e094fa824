#!/bin/bash
# Row-length / density / column laws beyond the named stand-ins, every kernel family + what adaptive settles on.
# Runs ON THE GPU BOX: writes gpurun_out/robustness.txt (progress goes there, so the run never looks hung).
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/robustness.txt
mkdir -p $R/gpurun_out; : > $O
run() { echo "\$ ${ROWLAW_M:+ROWLAW_M=$ROWLAW_M }${ROWLAW_COLS:+ROWLAW_COLS=$ROWLAW_COLS }$*" >> $O; timeout -k 10 200 "$@" 2>&1 | grep -v amdgpu.ids >> $O; echo >> $O; }
run python3 $R/tools/halves_bench.py 12 8
run python3 $R/tools/halves_bench.py 40 5
run python3 $R/tools/halves_bench.py 60 20
run python3 $R/tools/halves_bench.py 60 20 5000
run python3 $R/tools/halves_bench.py 30 10 64
run python3 $R/tools/rowlaw_bench.py lognormal 0.5
run python3 $R/tools/rowlaw_bench.py lognormal 1.0
run python3 $R/tools/rowlaw_bench.py lognormal 1.5
run python3 $R/tools/rowlaw_bench.py spikes 2000 3000
run python3 $R/tools/rowlaw_bench.py spikes 7500 600 5
run python3 $R/tools/rowlaw_bench.py spikes 30000 300 5
run python3 $R/tools/rowlaw_bench.py empty 0.5
ROWLAW_M=8000000 run python3 $R/tools/rowlaw_bench.py empty 0.97
ROWLAW_M=20000 run python3 $R/tools/rowlaw_bench.py lognormal 0.3 5000
ROWLAW_M=300 run python3 $R/tools/rowlaw_bench.py lognormal 0.2 300000
ROWLAW_M=8 run python3 $R/tools/rowlaw_bench.py lognormal 0.01 20000000
ROWLAW_M=1 run python3 $R/tools/rowlaw_bench.py lognormal 0.01 100000000
ROWLAW_M=50000000 run python3 $R/tools/rowlaw_bench.py spikes 200 30 0
ROWLAW_COLS=clusters run python3 $R/tools/rowlaw_bench.py lognormal 0.3
ROWLAW_COLS=uniform run python3 $R/tools/rowlaw_bench.py lognormal 0.3
run python3 $R/tools/unaligned_bench.py
run python3 $R/tools/host_overhead.py
tail -3 $O
