#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py -x -q -m gpu -k "overflow_kernel_beside or graph_capturable or concurrent" > gpurun_out/ovb_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -12 gpurun_out/ovb_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/lab_ab1.py "overflow_beside=1" "overflow_beside=0" "overflow_beside=1" ragged @rounds=3 2>&1 | grep -v amdgpu | cut -c1-300
timeout -k 10 300 python tools/lab_powerlaw.py 2>&1 | grep -v amdgpu | tail -8 | cut -c1-300
