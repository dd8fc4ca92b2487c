#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
P=gpurun_out/prof
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -m gpu -k "csc" > gpurun_out/csc_tests.log 2>&1
rc=$?; echo "csc tests rc=$rc"; tail -2 gpurun_out/csc_tests.log
[ $rc -ne 0 ] && exit 1
run() { local name=$1; shift; rm -rf $P/$name; timeout -k 10 420 "$@" > $P/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 1 $P/$name.log | cut -c1-300; [ $rc -ge 124 ] && exit $rc; }
run stats4 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats4 -o b -- python3 bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline
run fetch4 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch4 -o b -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
run write4 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write4 -o b -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
timeout -k 10 300 python bench.py --config 4 > gpurun_out/bench4.log 2>&1; echo "bench4 rc=$?"; tail -1 gpurun_out/bench4.log | cut -c1-400
