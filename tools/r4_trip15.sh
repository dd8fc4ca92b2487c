#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_csr_spmv.py tests/test_gpu_csr_fuzz.py -x -q > $O/t15_tests.log 2>&1; rc=$?; tail -n 3 $O/t15_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/t15_tests.log | head -20; exit $rc; }
for v in default sb4; do
  unset SPAL_HIP_LIB; [ $v = sb4 ] && export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/sb4/libspal_hip.so
  echo "== $v: config 2"; timeout -k 10 300 python tools/lab.py shard "slide_on=0,nt_store=0" "slide_on=1,nt_store=0" 2>&1 | grep -v amdgpu.ids | tee $O/t15_shard_$v.txt | tail -n 3
  echo "== $v: config 3, one super-tile per workgroup / sliding"; timeout -k 10 300 python tools/lab.py ab1 "slide_on=0" "slide_on=1" 2>&1 | grep -v amdgpu.ids | tee $O/t15_c3_$v.txt | tail -n 2
  echo "== $v: ragged rows"; timeout -k 10 300 python tools/lab.py ab1 "slide_on=0" "slide_on=1" ragged 2>&1 | grep -v amdgpu.ids | tee $O/t15_ragged_$v.txt | tail -n 2
done
exit 0
