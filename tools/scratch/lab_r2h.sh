#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 500 python tools/lab_place.py tune > gpurun_out/place_tune.log 2>&1; rc=$?; echo "place tune rc=$rc"; cat gpurun_out/place_tune.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 500 python bench.py --steps 100 --warmup 10 > gpurun_out/bench3.log 2>&1; rc=$?; echo "bench rc=$rc"; tail -n 3 gpurun_out/bench3.log
