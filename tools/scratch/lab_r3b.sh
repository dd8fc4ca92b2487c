#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu -k "csc" > gpurun_out/csc_tests.log 2>&1
rc=$?; echo "csc tests rc=$rc"; tail -25 gpurun_out/csc_tests.log
[ $rc -ne 0 ] && exit 1
for f in 0 2 1 0; do
  timeout -k 10 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --opt flush=$f > gpurun_out/b4_flush$f.log 2>&1
  echo "flush=$f rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b4_flush$f.log) $(grep -o '"flush": "[a-z_]*"' gpurun_out/b4_flush$f.log | head -1)"
done
