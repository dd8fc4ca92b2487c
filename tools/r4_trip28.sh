#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_csr_spmv.py tests/test_gpu_csr_fuzz.py -x -q > $O/t28_tests.log 2>&1; rc=$?; tail -n 3 $O/t28_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert\|error" $O/t28_tests.log | head -20; exit $rc; }
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/slstamps/libspal_hip.so timeout -k 10 300 python tools/lab.py shard "slide_on=1,slide_even=1" "slide_on=1,slide_even=0" @rounds=2 > $O/t28_stamps.log 2>&1
grep "slide stamps" $O/t28_stamps.log | tail -n 4 | cut -c1-400
timeout -k 10 300 python tools/lab.py shard "slide_on=1,slide_even=1" "slide_on=1,slide_even=0" "slide_on=0" @rounds=7 2>&1 | grep -v amdgpu.ids | tee $O/t28_shard.txt | tail -n 4
timeout -k 10 300 python tools/lab.py ab1 "slide_even=1" "slide_even=0" 2>&1 | grep -v amdgpu.ids | tee $O/t28_c3.txt | tail -n 2
timeout -k 10 300 python tools/lab.py ab1 "slide_even=1" "slide_even=0" ragged 2>&1 | grep -v amdgpu.ids | tee $O/t28_ragged.txt | tail -n 2
exit 0
