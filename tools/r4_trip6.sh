#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/t6_coo.log 2>&1; rc=$?; tail -n 3 $O/t6_coo.log; [ $rc -ne 0 ] && exit $rc
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/stamps/libspal_hip.so timeout -k 10 200 python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline > $O/t6_stamps.log 2>&1
grep "spal coo stamps" $O/t6_stamps.log | tail -n 1
for v in default lb7 lb5 norowsort; do
  unset SPAL_HIP_LIB SPAL_COO_NO_ROWSORT
  case $v in lb7|lb5) export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so;; norowsort) export SPAL_COO_NO_ROWSORT=1;; esac
  timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t6_b5_${v}.log 2>&1; rc=$?
  python - <<PY
import json
l=[x for x in open("$O/t6_b5_${v}.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v", d.get("ms_per_step"), d.get("product_plan_ms"))
PY
  [ $rc -ne 0 ] && exit $rc
done
exit 0
