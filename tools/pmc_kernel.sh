#!/bin/bash
# counters of one kernel of a bench command (rocprofv3 --pmc, one counter per pass, nothing else beside it):
#   bash tools/pmc_kernel.sh <tag> <kernel name substring> "<counters>" -- <bench args ...>
set -u
export TMPDIR=/tmp
TAG=$1; KERNEL=$2; COUNTERS=$3; shift 3; [ "$1" = "--" ] && shift
D=gpurun_out/r4/pmc_$TAG; mkdir -p $D
O=gpurun_out/r4/pmc_$TAG.txt
echo "# rocprofv3 --pmc <counter> -- python3 bench.py $* ; kernel *$KERNEL*; mean per launch over the launches of the pass" > $O
for c in $COUNTERS; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $D/$c -o b -- python3 bench.py "$@" > $D/$c.log 2>&1 || { echo "$c failed" >> $O; continue; }
  python3 - $c "$KERNEL" $D >> $O <<'P'
import csv,sys
c,k,d=sys.argv[1:4]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f"{d}/{c}/b_counter_collection.csv")) if k in r["Kernel_Name"]]
print(f"{c:32s} launches {len(v):3d}  mean {sum(v)/max(1,len(v)):.6g}")
P
done
cat $O
