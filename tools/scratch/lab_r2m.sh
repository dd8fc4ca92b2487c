#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py -x -q -m gpu -k "wide_bands or sliding or stream_global" > gpurun_out/pytest_panel.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 12 gpurun_out/pytest_panel.log
[ $rc -ne 0 ] && exit 1
for w in 8192 16384 32768 65536; do
timeout -k 10 500 python tools/lab_ab1.py "panel_on=1" "panel_on=0" @window=$w @rounds=3 > gpurun_out/ab1_panel_$w.log 2>&1; rc=$?; echo "ab1 W=$w rc=$rc"; cat gpurun_out/ab1_panel_$w.log
[ $rc -ge 124 ] && exit $rc
done
