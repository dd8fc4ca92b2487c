#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
P=gpurun_out/prof
run() { local name=$1; shift; rm -rf $P/$name; timeout -k 10 420 "$@" > $P/$name.log 2>&1; local rc=$?; echo "== $name rc=$rc"; tail -n 2 $P/$name.log | cut -c1-200; [ $rc -ge 124 ] && exit $rc; }
run stats5 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats5 -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline
run fetch5 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $P/fetch5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
run write5 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $P/write5 -o b -- python3 bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof/stats5/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    print(f"{r['Name'][:60]:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
