#!/bin/bash
set -u
mkdir -p gpurun_out
for v in csc2048 csc4096; do
 for f in 0 1; do
  export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so
  timeout -k 10 300 python bench.py --config 4 --steps 200 --warmup 20 --no-cpu-baseline --opt flush=$f > gpurun_out/b4_${v}_f$f.log 2>&1; rc=$?
  echo "== $v flush=$f rc=$rc"; tail -n 1 gpurun_out/b4_${v}_f$f.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['transposed_route']['ms_per_step'], d['transposed_route']['agrees_with_scatter'])"
  [ $rc -ge 124 ] && exit $rc
 done
done
