#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu > gpurun_out/coo_tests.log 2>&1
rc=$?; echo "coo tests rc=$rc"; tail -5 gpurun_out/coo_tests.log
[ $rc -ne 0 ] && exit 1
SPAL_COO_DEBUG=1 timeout -k 10 400 python bench.py --config 5 --steps 20 --warmup 3 > gpurun_out/bench5.log 2>&1
echo "bench5 rc=$?"; tail -4 gpurun_out/bench5.log | cut -c1-1500
