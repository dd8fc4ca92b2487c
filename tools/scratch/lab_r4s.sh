#!/bin/bash
set -u
SPAL_FUZZ_SEEDS=120 timeout -k 10 900 python -m pytest tests/test_gpu_csc_coo.py tests/test_gpu_csr_fuzz.py -x -q -m gpu > gpurun_out/soak2.log 2>&1
echo "soak rc=$?"; tail -1 gpurun_out/soak2.log
timeout -k 10 300 python tools/lab_coo_stress.py 60 2>&1 | grep -v amdgpu | tail -1
timeout -k 10 900 python -m pytest tests -q -m gpu > gpurun_out/pytest_gpu.log 2>&1
echo "full rc=$?"; tail -1 gpurun_out/pytest_gpu.log
