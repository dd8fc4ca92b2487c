#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
SPAL_FUZZ_SEEDS=100 timeout -k 10 1100 python -m pytest tests/test_gpu_csc_coo.py -x -q > $O/soak_coo.log 2>&1; rc=$?; tail -n 3 $O/soak_coo.log; [ $rc -ne 0 ] && { grep -n "Error\|assert\|seed" $O/soak_coo.log | head -20; exit $rc; }
exit 0
