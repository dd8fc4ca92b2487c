#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 25 gpurun_out/pytest_gpu.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 500 python tools/lab_ab1.py "slide=0" "slide=-1,slide_on=0" "slide=-1,slide_on=1" "slide=-1,slide_on=1,uniform_rows=0" "slide=-1,slide_on=1,nt_store=1" @rounds=4 > gpurun_out/ab1_slide.log 2>&1; rc=$?; echo "ab1 rc=$rc"; cat gpurun_out/ab1_slide.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 500 python tools/lab_ab1.py "slide=0" "slide=-1,slide_on=1" ragged @rounds=3 > gpurun_out/ab1_slide_ragged.log 2>&1; rc=$?; echo "ab1 ragged rc=$rc"; cat gpurun_out/ab1_slide_ragged.log
