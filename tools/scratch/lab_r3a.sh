#!/bin/bash
set -u
mkdir -p gpurun_out
s=$(date +%s)
timeout -k 10 900 python bench.py > gpurun_out/bench_default.log 2>&1
echo "default bench rc=$? in $(( $(date +%s) - s )) s"
tail -1 gpurun_out/bench_default.log | python3 -c "
import sys,json
j=json.loads(sys.stdin.readline())
print({k:j[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','vs_baseline','dtype','data')})
print('roofline',j['roofline'])
print('cpu_baseline',j['cpu_baseline'])
print([k for k in j])
"
