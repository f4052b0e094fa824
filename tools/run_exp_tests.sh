#!/bin/bash
# the col16 / parity tests against the EXPERIMENTAL build (spmv_acc_amd/lib_exp), by swapping libraries inside the GPU box's copy of the tree
set -e
cd $GRAFT_REPO_ROOT
cp spmv_acc_amd/lib/libspmv_acc.so /tmp/shipped.so
cp spmv_acc_amd/lib_exp/libspmv_acc.so spmv_acc_amd/lib/libspmv_acc.so
timeout -k 10 400 python -m pytest tests/test_gpu_col16.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/exp_tests.txt 2>&1 || true
tail -n 3 gpurun_out/exp_tests.txt
cp /tmp/shipped.so spmv_acc_amd/lib/libspmv_acc.so
