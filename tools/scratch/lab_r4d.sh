#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29811 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-ceiling --same-device --backend gloo > gpurun_out/bench_gloo2.log 2>&1
echo "bench 2 ranks gloo same device rc=$?"; tail -1 gpurun_out/bench_gloo2.log | cut -c1-1200
timeout -k 10 300 python bench.py --host mg --gpus 2 --devices 0,0 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_mg2.log 2>&1
echo "bench --host mg 2 virtual shards rc=$?"; tail -1 gpurun_out/bench_mg2.log | cut -c1-1000
