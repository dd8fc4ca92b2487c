#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 4 gpurun_out/pytest_gpu.log
[ $rc -ge 124 ] && exit $rc
tools/micro/stream_ceiling --quick 10000000 14 | tee gpurun_out/quick1.log
timeout -k 10 500 python tools/lab_ab1.py "slide_run=0" "slide_run=4" "slide_run=8" "slide_run=16" "slide_run=32" "slide_run=16,persistent_blocks=1024" "slide_on=0" @rounds=3 > gpurun_out/ab1_run.log 2>&1; rc=$?; echo "ab1 rc=$rc"; cat gpurun_out/ab1_run.log
