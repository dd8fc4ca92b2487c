#!/bin/bash
for o in "cols_per_block=4096" "cols_per_block=4096 --opt flush=2" "cols_per_block=2048 --opt flush=2" "cols_per_block=1024 --opt flush=2" "cols_per_block=2048 --opt flush=1"; do
  timeout -k 10 300 python bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline --opt $o > gpurun_out/b4x.log 2>&1
  echo "$o: $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/b4x.log | head -1) $(grep -o '"flush": "[a-z_]*"' gpurun_out/b4x.log | head -1)"
done
