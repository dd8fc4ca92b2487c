#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py -x -q -m gpu -k "wide_bands or sliding or stream_global or patchwork" > gpurun_out/pytest_panel.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 12 gpurun_out/pytest_panel.log
[ $rc -ne 0 ] && exit 1
tools/micro/stream_ceiling --quick 10000000 14
timeout -k 10 500 python tools/lab_ab1.py "slide_on=1" "slide_on=0" @rounds=3 > gpurun_out/ab1_slide2.log 2>&1; echo "ab1 rc=$?"; cat gpurun_out/ab1_slide2.log
for w in 16384 32768 65536; do
timeout -k 10 500 python tools/lab_ab1.py "panel_on=1,panel_pages=320" "panel_on=0" @window=$w @rounds=3 > gpurun_out/ab1_panel2_$w.log 2>&1; rc=$?; echo "ab1 W=$w rc=$rc"; cat gpurun_out/ab1_panel2_$w.log
[ $rc -ge 124 ] && exit $rc
done
