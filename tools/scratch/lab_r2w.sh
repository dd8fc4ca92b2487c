#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu > gpurun_out/coo_tests.log 2>&1
rc=$?; echo "coo tests rc=$rc"; tail -2 gpurun_out/coo_tests.log
[ $rc -ne 0 ] && exit 1
for v in "$@"; do
  if [ $v = default ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  SPAL_COO_DEBUG=1 timeout -k 10 300 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b5_$v.log 2>&1
  rc=$?
  echo "$v rc=$rc $(grep -m1 'spal coo' gpurun_out/b5_$v.log) $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b5_$v.log)"
  [ $rc -ge 124 ] && exit 1
done
exit 0
