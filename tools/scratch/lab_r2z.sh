#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29801 tests/dist_nccl_worker.py > gpurun_out/nccl1_worker.log 2>&1
rc=$?; echo "nccl worker (1 rank) rc=$rc"; tail -3 gpurun_out/nccl1_worker.log | cut -c1-400
[ $rc -ge 124 ] && exit 1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29802 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-ceiling > gpurun_out/bench_torchrun1.log 2>&1
rc=$?; echo "bench under torchrun (1 rank) rc=$rc"; tail -2 gpurun_out/bench_torchrun1.log | cut -c1-600
[ $rc -ge 124 ] && exit 1
timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29803 bench.py --gpus 2 --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-ceiling --same-device --backend nccl > gpurun_out/bench_nccl_same_device.log 2>&1
rc=$?; echo "bench, 2 nccl ranks on one GPU rc=$rc"; tail -4 gpurun_out/bench_nccl_same_device.log | cut -c1-600
exit 0
