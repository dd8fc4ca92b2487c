#!/bin/bash
set -u
export TMPDIR=/tmp
P=gpurun_out/prof
mkdir -p $P
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $P/uni_$c
  timeout -k 10 420 rocprofv3 --pmc $c --output-format csv -d $P/uni_$c -o b -- python3 bench.py --config 2 --dist uniform --steps 10 --warmup 3 --no-cpu-baseline --no-ceiling --copies 1 > $P/uni_$c.log 2>&1
  echo "$c rc=$?"
done
python3 - <<'PY'
import csv,glob,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f'gpurun_out/prof/uni_{c}/**/*counter_collection.csv',recursive=True)[0]
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'csr_spmv' in r['Kernel_Name']: acc[r['Kernel_Name'][:50]].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(c,k,len(v),sum(v)/len(v))
PY
tail -1 gpurun_out/prof/uni_FETCH_SIZE.log | cut -c1-300
