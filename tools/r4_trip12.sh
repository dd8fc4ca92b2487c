#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for rep in 1 2 3; do
for v in default prev; do
  unset SPAL_HIP_LIB
  case $v in prev) export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/prev/libspal_hip.so;; esac
  timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t12_b5_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/t12_b5_$v.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v:", d.get("ms_per_step"), d.get("product_plan_ms"))
PY
done
done
for v in default prev; do
  unset SPAL_HIP_LIB
  case $v in prev) export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/prev/libspal_hip.so;; esac
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5g_$v -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t12_p5_$v.log 2>&1
  python - <<PY
import csv
print("== $v")
for r in list(csv.DictReader(open("$O/stats5g_$v/b_kernel_stats.csv")))[:18]:
    if any(k in r["Name"] for k in ("radix","scan","coo_group","group_offsets","groups_check","rows_b")): print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"])
PY
done
exit 0
