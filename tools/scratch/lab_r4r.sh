#!/bin/bash
set -u
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu -k "coo" > gpurun_out/coo_tests.log 2>&1
rc=$?; echo "coo tests rc=$rc"; tail -1 gpurun_out/coo_tests.log
[ $rc -ne 0 ] && exit 1
for v in default coo_w7 default coo_w7; do
  if [ $v = default ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  timeout -k 10 300 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/b5_$v.log 2>&1
  echo "$v rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b5_$v.log) $(grep -o '"gpu_assembly_equals_cpu_bit_for_bit": [a-z]*' gpurun_out/b5_$v.log)"
done
