#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py -x -q > $O/t14_tests.log 2>&1; rc=$?; tail -n 3 $O/t14_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/t14_tests.log | head -20; exit $rc; }
echo "== default build (first window in one batch)"; timeout -k 10 300 python tools/lab.py shard 2>&1 | tee $O/t14_shard_default.txt | tail -n 7
echo "== first window four vectors at a time (round 3)"; SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/fb4/libspal_hip.so timeout -k 10 300 python tools/lab.py shard "slide_on=1,nt_store=0" "slide_on=1,nt_store=0,arith_bounds=0" 2>&1 | tee $O/t14_shard_fb4.txt | tail -n 4
echo "== grids"; timeout -k 10 300 python tools/lab.py shard "slide_on=1,persistent_blocks=512" "slide_on=1,persistent_blocks=256" "slide_on=1,persistent_blocks=384" "slide_on=1,slide_run=2" "slide_on=1,slide_run=4" 2>&1 | tee $O/t14_shard_grids.txt | tail -n 7
exit 0
