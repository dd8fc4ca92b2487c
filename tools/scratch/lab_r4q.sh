#!/bin/bash
set -u
SPAL_FUZZ_SEEDS=120 timeout -k 10 900 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu -k "coo" > gpurun_out/coo_soak.log 2>&1
rc=$?; echo "coo tests rc=$rc"; tail -2 gpurun_out/coo_soak.log
[ $rc -ne 0 ] && exit 1
for i in 1 2; do
SPAL_COO_DEBUG=1 timeout -k 10 400 python bench.py --config 5 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench5.log 2>&1
echo "bench5 rc=$? $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/bench5.log)"
done
timeout -k 10 300 python tools/lab_coo_stress.py 60 2>&1 | grep -v amdgpu | tail -1
