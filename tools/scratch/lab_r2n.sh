#!/bin/bash
set -u
mkdir -p gpurun_out
for v in main csc2048 csc4096; do
  if [ "$v" = main ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  timeout -k 10 300 python bench.py --config 4 --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/b4_$v.log 2>&1; rc=$?
  echo "== $v rc=$rc"; tail -n 1 gpurun_out/b4_$v.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['config']['plan'])"
  [ $rc -ge 124 ] && exit $rc
done
unset SPAL_HIP_LIB
timeout -k 10 300 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu > gpurun_out/pytest_csc.log 2>&1; echo "tests rc=$?"; tail -n 3 gpurun_out/pytest_csc.log
timeout -k 10 300 python bench.py --config 5 --no-cpu-baseline > gpurun_out/b5n.log 2>&1; echo "b5 rc=$?"; tail -n 1 gpurun_out/b5n.log | cut -c1-400
