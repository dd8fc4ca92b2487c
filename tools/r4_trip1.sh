#!/bin/bash
# round 4, trip 1: the resident group kernel -- parity first, then time
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/t1_coo.log 2>&1; rc=$?; tail -n 3 $O/t1_coo.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t1_b5.log 2>&1; rc=$?; tail -n 1 $O/t1_b5.log | cut -c1-600; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5 -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t1_p5.log 2>&1; rc=$?; echo "prof rc=$rc"; [ $rc -ne 0 ] && exit $rc
timeout -k 10 120 tools/micro/wide_scatter > $O/wide_scatter.txt 2>&1; rc=$?; cat $O/wide_scatter.txt; [ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_cblock.py tests/test_gpu_csc_coo.py -x -q > $O/t1_rest.log 2>&1; rc=$?; tail -n 3 $O/t1_rest.log; exit $rc
