#!/bin/bash
# MatrixMarket files of a directory -> bin2 files, one `spmv-cli <f>.mtx -f mtx --convert-bin2 <out>/<f>.bin2` each: the batch mode of the
# reference's converter (tools/suitesparse-dl/conv/conv.go:58-84 reads a name,path CSV; here the directory is the list).
# usage: tools/convert_dir.sh <dir with *.mtx> <output dir>
set -u
DIR=${1:?usage: $0 <dir with *.mtx> <output dir>}; OUT=${2:?usage: $0 <dir with *.mtx> <output dir>}
BIN="$(cd "$(dirname "$0")/.." && pwd)/spmv_acc_amd/bin/spmv-cli"
[ -x "$BIN" ] || { echo "build first: make -C spmv_acc_amd/csrc" >&2; exit 2; }
mkdir -p "$OUT" || exit 2
rc=0
for f in "$DIR"/*.mtx; do
  [ -e "$f" ] || continue
  "$BIN" "$f" -f mtx --convert-bin2 "$OUT/$(basename "${f%.mtx}").bin2" || { echo "conversion failed: $f" >&2; rc=1; }
done
exit $rc
