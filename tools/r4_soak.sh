#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
SPAL_FUZZ_SEEDS=160 timeout -k 10 1000 python -m pytest tests/test_gpu_csr_fuzz.py -x -q > $O/soak_csr.log 2>&1; rc=$?; tail -n 3 $O/soak_csr.log; [ $rc -ne 0 ] && { grep -n "Error\|assert\|seed" $O/soak_csr.log | head -20; exit $rc; }
exit 0
