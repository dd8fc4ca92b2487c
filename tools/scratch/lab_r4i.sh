#!/bin/bash
set -u
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py -x -q -m gpu -k "wide_bands or panel" 2>&1 | tail -4
for w in 16384 32768; do
  timeout -k 10 400 python tools/lab_ab1.py "panel_on=1" "panel_on=0" @window=$w @rounds=2 2>&1 | grep -v amdgpu | cut -c1-250
done
