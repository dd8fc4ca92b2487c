#!/bin/bash
set -u
export TMPDIR=/tmp
P=gpurun_out/prof
mkdir -p $P
rm -rf $P/stats_ragged
timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $P/stats_ragged -o b -- python3 bench.py --dist ragged --steps 50 --warmup 5 --no-cpu-baseline --no-ceiling > $P/stats_ragged.log 2>&1
echo "rc=$?"; tail -1 $P/stats_ragged.log | cut -c1-300
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof/stats_ragged/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs'])>1e6: print(f"   {r['Name'][:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
