#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py tests/test_gpu_cblock.py -x -q > $O/t11_tests.log 2>&1; rc=$?; tail -n 3 $O/t11_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/t11_tests.log | head -20; exit $rc; }
for v in default loop lb8loop default; do
  unset SPAL_COO_LOOP_RANKS SPAL_HIP_LIB
  case $v in loop) export SPAL_COO_LOOP_RANKS=1;; lb8loop) export SPAL_COO_LOOP_RANKS=1 SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/lb8/libspal_hip.so;; esac
  timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t11_b5_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/t11_b5_$v.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v:", d.get("ms_per_step"), d.get("product_plan_ms"))
PY
done
unset SPAL_COO_LOOP_RANKS SPAL_HIP_LIB
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5f -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t11_p5.log 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats5f/b_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'coo_group_sort' in r['Kernel_Name']]
start=idx[-3]+1; last=idx[-2]
t0=int(rows[start]['Start_Timestamp']); prev=None
for r in rows[start:last+4]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us dur {(e-s)/1e3:8.1f} gap {((s-prev)/1e3 if prev else 0):7.1f} {r['Kernel_Name'][:60]}")
    prev=e
PY
exit 0
