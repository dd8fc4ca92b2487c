#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_mg.py -x -q -m gpu > gpurun_out/pytest_mg.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 25 gpurun_out/pytest_mg.log | cut -c1-300
[ $rc -ge 124 ] && exit $rc
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --same-device --config 2 > gpurun_out/rehearse2.log 2>&1; rc=$?; echo "rehearse2 rc=$rc"; tail -n 3 gpurun_out/rehearse2.log | cut -c1-2500
