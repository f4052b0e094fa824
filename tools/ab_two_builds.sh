#!/bin/bash
# A/B of two builds of the library on ONE box: spmv_acc_amd/lib (shipped) against spmv_acc_amd/lib_exp (make ... OBJ_DIR=build_exp OUT_DIR=../lib_exp EXTRA=-D...)
set -e
cd $GRAFT_REPO_ROOT
run() { timeout -k 10 400 python -u tools/ab_steady.py adaptive Hardesty3,RM07R,boneS10,Bump_2911,largebasis,af_shell10 ";stream_plain=0" 2>&1 | grep "max difference"; timeout -k 10 200 python -u tools/ab_steady.py flat Hardesty3,RM07R ";stream_plain=0" 2>&1 | grep "max difference"; }
echo "== base"; run
cp spmv_acc_amd/lib/libspmv_acc.so /tmp/base.so
# whatever happens below (set -e, a failing test), the shipped library comes back
trap 'cp /tmp/base.so spmv_acc_amd/lib/libspmv_acc.so' EXIT
cp spmv_acc_amd/lib_exp/libspmv_acc.so spmv_acc_amd/lib/libspmv_acc.so
echo "== experimental build"; python -m pytest tests/test_gpu_parity.py -x -q -k "parity_all_strategies or randomised_shapes" 2>&1 | tail -1; run
cp /tmp/base.so spmv_acc_amd/lib/libspmv_acc.so
echo "== base again"; timeout -k 10 400 python -u tools/ab_steady.py adaptive Hardesty3,RM07R ";stream_plain=0" 2>&1 | grep "max difference"
