#!/bin/bash
# Run spmv-cli over every matrix file of a directory (what the reference's examples/*-batch.sh job scripts do for a fixed
# list on a Slurm cluster), picking the reader from the suffix: *.csr (5-line text), *.bin2 / *.bin (binary CSR), *.mtx
# (MatrixMarket).  Extra arguments go to spmv-cli, e.g.
#     tools/run_matrix_dir.sh ./large-data-set --benchmark            # one PERFORMANCE,... CSV row per matrix and strategy
#     SPMV_ACC_KERNEL_STRATEGY=flat tools/run_matrix_dir.sh ./large-data-set
# usage: tools/run_matrix_dir.sh <dir> [spmv-cli options...]
set -u
DIR=${1:?usage: $0 <dir> [spmv-cli options...]}; shift
BIN="$(cd "$(dirname "$0")/.." && pwd)/spmv_acc_amd/bin/spmv-cli"
[ -x "$BIN" ] || { echo "build first: make -C spmv_acc_amd/csrc" >&2; exit 2; }
rc=0
for f in "$DIR"/*; do
  case "$f" in
    *.csr) fmt=csr ;;
    *.bin2|*.bin) fmt=bin2 ;;
    *.mtx) fmt=mtx ;;
    *) continue ;;
  esac
  echo "== $(basename "$f")"
  "$BIN" "$f" -f "$fmt" "$@" || { echo "spmv-cli failed on $f" >&2; rc=1; }
done
exit $rc
