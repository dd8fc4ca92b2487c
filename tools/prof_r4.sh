#!/bin/bash
# round-4 profiles: per-kernel stats and HBM counters of the bench commands (rocprofv3; the counter passes are runs of
# their own, --pmc never beside a trace), summarised into profiles/r04 by tools/profile_summary4.py
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof4
P=gpurun_out/prof4
run() { local name=$1; shift; timeout -k 10 420 "$@" > $P/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 1 $P/$name.log | cut -c1-160; [ $rc -ge 124 ] && exit $rc; }
B="--no-cpu-baseline --no-ceiling --no-other-configs"
run stats3 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats3 -o b -- python3 bench.py --steps 100 --warmup 10 $B
run fetch3 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch3 -o b -- python3 bench.py --steps 20 --warmup 5 $B
run write3 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write3 -o b -- python3 bench.py --steps 20 --warmup 5 $B
run stats3f32 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats3f32 -o b -- python3 bench.py --dtype f32 --steps 100 --warmup 10 $B
run fetch3f32 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch3f32 -o b -- python3 bench.py --dtype f32 --steps 20 --warmup 5 $B
run write3f32 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write3f32 -o b -- python3 bench.py --dtype f32 --steps 20 --warmup 5 $B
run stats2 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats2 -o b -- python3 bench.py --config 2 --steps 100 --warmup 10 $B
run fetch2 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch2 -o b -- python3 bench.py --config 2 --steps 20 --warmup 5 $B --copies 1
run write2 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write2 -o b -- python3 bench.py --config 2 --steps 20 --warmup 5 $B --copies 1
run stats2u rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats2u -o b -- python3 bench.py --config 2 --dist uniform --steps 100 --warmup 10 $B
run fetch2u rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch2u -o b -- python3 bench.py --config 2 --dist uniform --steps 20 --warmup 5 $B --copies 1
run stats4 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats4 -o b -- python3 bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline
run fetch4 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch4 -o b -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
run write4 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write4 -o b -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
run stats5 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats5 -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
run fetch5 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
run write5 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
run stats1 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats1 -o b -- python3 bench.py --config 1 --steps 100 --warmup 10 --no-cpu-baseline
find $P -name "*.csv" | wc -l
