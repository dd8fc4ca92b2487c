#!/bin/bash
# round-2 profiles: per-kernel stats and HBM counters of the bench commands (rocprofv3), copied to profiles/r02 by hand
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
P=gpurun_out/prof
run() { local name=$1; shift; timeout -k 10 420 "$@" > $P/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 2 $P/$name.log | cut -c1-200; [ $rc -ge 124 ] && exit $rc; }
B3="python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-ceiling"
run stats3 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats3 -o b -- $B3
run fetch3 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch3 -o b -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ceiling
run write3 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write3 -o b -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-ceiling
run sq3 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/sq3 -o b -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-ceiling
run stats2 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats2 -o b -- python3 bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline --no-ceiling
run fetch2 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch2 -o b -- python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-ceiling --copies 1
run write2 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write2 -o b -- python3 bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-ceiling --copies 1
run stats4 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats4 -o b -- python3 bench.py --config 4 --steps 100 --warmup 10 --no-cpu-baseline
run fetch4 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch4 -o b -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
run write4 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write4 -o b -- python3 bench.py --config 4 --steps 10 --warmup 3 --no-cpu-baseline --copies 1
run stats5 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats5 -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
run fetch5 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
run write5 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
# LDS bank conflicts: the one-super-tile kernel as shipped vs the conflict-free (wrong-result) ablation build
export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/diag/libspal_hip.so
run sq3_plain rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/sq3_plain -o b -- python3 tools/lab.py ab1 "diag=0,slide_on=0" @rounds=1
run sq3_noconf rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $P/sq3_noconf -o b -- python3 tools/lab.py ab1 "diag=256,slide_on=0" @rounds=1
unset SPAL_HIP_LIB
find $P -name "*.csv" | head -40
