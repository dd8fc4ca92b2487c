#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_csr_spmv.py tests/test_gpu_csr_fuzz.py -x -q > $O/t29_tests.log 2>&1; rc=$?; tail -n 3 $O/t29_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert\|error" $O/t29_tests.log | head -20; exit $rc; }
timeout -k 10 300 python tools/lab.py powerlaw quick 2>&1 | grep -v amdgpu.ids | tee $O/t29_powerlaw.txt | grep "row_split': -1}\|row_split': 0}\|power-law"
exit 0
