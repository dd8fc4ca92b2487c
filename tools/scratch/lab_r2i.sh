#!/bin/bash
set -u
mkdir -p gpurun_out
tools/micro/stream_ceiling --quick 10000000 14 | tee gpurun_out/quick1.log
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/diag/libspal_hip.so timeout -k 10 500 python tools/lab_ab1.py "diag=0,slide_on=1" "diag=1024,slide_on=1" "diag=2048,slide_on=1" "diag=6144,slide_on=1" "diag=0,slide_on=0" "diag=512,slide_on=0" "diag=1536,slide_on=0" @rounds=3 > gpurun_out/ab1_sdiag.log 2>&1; rc=$?; echo "ab1 rc=$rc"; cat gpurun_out/ab1_sdiag.log
[ $rc -ge 124 ] && exit $rc
tools/micro/stream_ceiling --quick 10000000 14 | tee gpurun_out/quick2.log
