#!/bin/bash
set -u
mkdir -p gpurun_out
for v in default csc_u1 csc_u2 csc_u3 default; do
  if [ $v = default ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  timeout -k 10 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/b4_$v.log 2>&1
  rc=$?
  echo "$v rc=$rc $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b4_$v.log | head -1) $(grep -o '"agrees_with_scatter": [a-z]*' gpurun_out/b4_$v.log)"
  [ $rc -ge 124 ] && exit 1
done
exit 0
