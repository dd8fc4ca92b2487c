#!/bin/bash
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
P=gpurun_out/prof
for v in default coo_skip; do
  if [ $v = default ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  rm -rf $P/st_$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $P/st_$v -o b -- python3 tools/lab_coo_once.py > $P/st_$v.log 2>&1
  echo "$v rc=$?"
  python3 - $P/st_$v <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if float(r['TotalDurationNs'])>2e5: print(f"   {r['Name'][:50]:50s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
done
