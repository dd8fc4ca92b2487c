#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for nr in 0 1; do
 SPAL_COO_NO_ROWSORT=$nr SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/stamps/libspal_hip.so timeout -k 10 200 python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline > $O/t7_stamps_$nr.log 2>&1
 echo "NO_ROWSORT=$nr"; grep "spal coo stamps" $O/t7_stamps_$nr.log | tail -n 1
done
exit 0
