#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
for tk in 8 0; do
  SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/stamps/libspal_hip.so SPAL_COO_TICKET=$tk timeout -k 10 200 python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline > $O/t5_stamps_$tk.log 2>&1; rc=$?
  echo "ticket $tk"; grep "spal coo stamps" $O/t5_stamps_$tk.log | tail -n 2
  [ $rc -ne 0 ] && exit $rc
done
for v in default lb8; do
  if [ $v = default ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t5_b5_${v}_$rep.log 2>&1; rc=$?
    python - <<PY
import json
l=[x for x in open("$O/t5_b5_${v}_$rep.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v", d.get("ms_per_step"), d.get("product_plan_ms"), (d.get("spmv_on_result") or {}).get("first_product_ms"))
PY
    [ $rc -ne 0 ] && exit $rc
  done
done
