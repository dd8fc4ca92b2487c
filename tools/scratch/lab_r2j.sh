#!/bin/bash
set -u
mkdir -p gpurun_out
tools/micro/stream_ceiling --quick 10000000 14 | tee gpurun_out/quick1.log
timeout -k 10 500 python tools/lab_ab1.py "persistent_blocks=0" "persistent_blocks=768" "persistent_blocks=1024" "persistent_blocks=1536" "persistent_blocks=2048" "persistent_blocks=4096" "slide_on=0" @rounds=3 > gpurun_out/ab1_grid.log 2>&1; rc=$?; echo "ab1 rc=$rc"; cat gpurun_out/ab1_grid.log
