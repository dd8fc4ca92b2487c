#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 800 python -m pytest tests/test_gpu_csr_spmv.py tests/test_gpu_csr_fuzz.py -x -q -m gpu > gpurun_out/split_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -14 gpurun_out/split_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 500 python tools/lab_ab1.py "split_tiles=1" "split_tiles=0" "split_tiles=1" ragged @rounds=3 2>&1 | grep -v amdgpu | cut -c1-300
