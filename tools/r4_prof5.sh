#!/bin/bash
# config 5's profile passes again (the rest of gpurun_out/prof4 stands): per-kernel stats, FETCH_SIZE, WRITE_SIZE -- each its own run
set -u
export TMPDIR=/tmp
P=gpurun_out/prof4; mkdir -p $P
run() { local name=$1; shift; rm -rf $P/$name; timeout -k 10 420 "$@" > $P/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 1 $P/$name.log | cut -c1-160; [ $rc -ge 124 ] && exit $rc; }
run stats5 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats5 -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
run fetch5 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
run write5 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
exit 0
