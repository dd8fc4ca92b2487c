#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py -x -q -m gpu -k "row_blocks or multi_gpu or autotune or device_path" > gpurun_out/rb_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -15 gpurun_out/rb_tests.log
[ $rc -ne 0 ] && exit 1
timeout -k 10 900 python tools/lab_huge.py 330000000 > gpurun_out/lab_huge_row_blocks.log 2>&1
echo "huge rc=$?"; cat gpurun_out/lab_huge_row_blocks.log | cut -c1-700
