#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 400 python tools/lab_place.py > gpurun_out/place.log 2>&1; rc=$?; echo "place rc=$rc"; cat gpurun_out/place.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python tools/lab_place.py prefetch=2 > gpurun_out/place_pf2.log 2>&1; rc=$?; echo "place pf2 rc=$rc"; cat gpurun_out/place_pf2.log
[ $rc -ge 124 ] && exit $rc
timeout -k 10 400 python tools/lab_ab1.py "uniform_rows=0,prefetch=1,persistent=0" "uniform_rows=1,prefetch=1,persistent=0" "uniform_rows=1,prefetch=2,persistent=0" "uniform_rows=0,prefetch=2,persistent=0" "uniform_rows=1,prefetch=1,persistent=1" "uniform_rows=0,prefetch=1,persistent=1" > gpurun_out/ab1_main.log 2>&1; rc=$?; echo "ab1 rc=$rc"; cat gpurun_out/ab1_main.log
[ $rc -ge 124 ] && exit $rc
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/diag/libspal_hip.so timeout -k 10 500 python tools/lab_ab1.py "diag=0,prefetch=1" "diag=256,prefetch=1" "diag=512,prefetch=1" "diag=1024,prefetch=1" "diag=1536,prefetch=1" "diag=0,prefetch=2" "diag=256,prefetch=2" "diag=512,prefetch=2" "diag=1536,prefetch=2" > gpurun_out/ab1_diag.log 2>&1; echo "ab1diag rc=$?"; cat gpurun_out/ab1_diag.log
