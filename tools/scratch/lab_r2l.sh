#!/bin/bash
set -u
mkdir -p gpurun_out
run() { local name=$1; shift; timeout -k 10 500 "$@" > gpurun_out/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 1 gpurun_out/$name.log | cut -c1-3000; [ $rc -ge 124 ] && exit $rc; }
run b3 python bench.py --steps 100 --warmup 10
run b2 python bench.py --config 2 --steps 200 --warmup 20
run b3ragged python bench.py --dist ragged --steps 50 --warmup 5 --no-cpu-baseline
run b3uniform python bench.py --dist uniform --steps 20 --warmup 3 --no-cpu-baseline
run b4 python bench.py --config 4 --steps 100 --warmup 10
run b5 python bench.py --config 5
run bmg python bench.py --host mg --devices 0,0,0,0 --config 2 --steps 50 --warmup 5
