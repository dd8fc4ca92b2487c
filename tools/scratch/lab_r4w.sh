#!/bin/bash
set -u
export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/coo_big/libspal_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu -k "coo and not geometries" > gpurun_out/coo_tests_big.log 2>&1
rc=$?; echo "coo tests (big groups) rc=$rc"; tail -1 gpurun_out/coo_tests_big.log
[ $rc -ne 0 ] && exit 1
for v in default coo_big default coo_big; do
  if [ $v = default ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  SPAL_COO_DEBUG=1 timeout -k 10 300 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b5_$v.log 2>&1
  echo "$v rc=$? $(grep -m1 'spal coo' gpurun_out/b5_$v.log | cut -c1-100) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b5_$v.log)"
done
