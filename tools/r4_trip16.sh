#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
: > $O/t16_shard.txt
for rep in 1 2 3 4; do
for v in default sb4 fb4; do
  unset SPAL_HIP_LIB; [ $v != default ] && export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so
  echo "== $v rep $rep" >> $O/t16_shard.txt
  timeout -k 10 300 python tools/lab.py shard "slide_on=0,nt_store=0" "slide_on=1,nt_store=0" @rounds=3 2>&1 | grep "median" >> $O/t16_shard.txt
done
done
python - <<PY
import re, collections, statistics
acc=collections.defaultdict(list); cur=None
for line in open("$O/t16_shard.txt"):
    if line.startswith("=="): cur=line.split()[1]; continue
    m=re.match(r"(\S+)\s+median\s+([\d.]+)", line)
    if m: acc[(cur, m.group(1))].append(float(m.group(2)))
for k,v in sorted(acc.items()): print(k, [round(x,2) for x in v], "median", round(statistics.median(v),2))
PY
exit 0
