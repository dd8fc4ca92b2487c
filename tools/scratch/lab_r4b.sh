#!/bin/bash
set -u
mkdir -p gpurun_out
SPAL_FUZZ_SEEDS=240 timeout -k 10 1000 python -m pytest tests/test_gpu_csr_fuzz.py tests/test_gpu_csc_coo.py -q -m gpu -x > gpurun_out/soak.log 2>&1
echo "soak rc=$?"; tail -5 gpurun_out/soak.log
