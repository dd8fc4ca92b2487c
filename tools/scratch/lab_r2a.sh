#!/bin/bash
# round-2 lab call A: copy ceiling, GPU tests, A/B of uniform rows / prefetch depth, ablations (diag build)
set -u
mkdir -p gpurun_out
timeout -k 10 300 tools/micro/stream_ceiling > gpurun_out/ceiling.log 2>&1; echo "ceiling rc=$?"
tail -n 40 gpurun_out/ceiling.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 5 gpurun_out/pytest_gpu.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 500 python tools/lab_ab.py "uniform_rows=0" "uniform_rows=1" "prefetch=2" "prefetch=2,uniform_rows=0" "persistent=1" "persistent=1,uniform_rows=0" > gpurun_out/ab_main.log 2>&1; rc=$?; echo "ab rc=$rc"; cat gpurun_out/ab_main.log
[ $rc -ge 124 ] && exit $rc
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/diag/libspal_hip.so timeout -k 10 500 python tools/lab_ab.py "diag=0" "diag=256" "diag=512" "diag=1024" "diag=1536" "diag=0,prefetch=2" "diag=1536,prefetch=2" > gpurun_out/ab_diag.log 2>&1; echo "abdiag rc=$?"; cat gpurun_out/ab_diag.log
