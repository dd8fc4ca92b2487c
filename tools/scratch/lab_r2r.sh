#!/bin/bash
set -u
for w in 16384 32768 65536; do
timeout -k 10 500 python tools/lab_ab1.py "panel_pages=320,panel_window=0" "panel_pages=320,panel_window=78" "panel_pages=320,panel_window=56" @window=$w @rounds=3 > gpurun_out/ab1_panel3_$w.log 2>&1; rc=$?; echo "ab1 W=$w rc=$rc"; cat gpurun_out/ab1_panel3_$w.log
[ $rc -ge 124 ] && exit $rc
done
