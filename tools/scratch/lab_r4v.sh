#!/bin/bash
SPAL_FUZZ_SEEDS=400 timeout -k 10 1100 python -m pytest tests/test_gpu_csr_fuzz.py -q -m gpu -x -p no:cacheprovider > gpurun_out/soak3.log 2>&1
echo "soak rc=$?"; tail -3 gpurun_out/soak3.log
