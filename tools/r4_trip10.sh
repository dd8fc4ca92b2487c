#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/t10_coo.log 2>&1; rc=$?; tail -n 3 $O/t10_coo.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/t10_coo.log | head -20; exit $rc; }
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/stamps/libspal_hip.so timeout -k 10 200 python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline > $O/t10_stamps.log 2>&1
grep "spal coo stamps" $O/t10_stamps.log | tail -n 1
for v in default loop default; do
  unset SPAL_COO_LOOP_RANKS
  case $v in loop) export SPAL_COO_LOOP_RANKS=1;; esac
  timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t10_b5_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/t10_b5_$v.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v:", d.get("ms_per_step"), d.get("product_plan_ms"), d.get("dtype"))
PY
done
unset SPAL_COO_LOOP_RANKS
timeout -k 10 200 python bench.py --config 5 --dtype f32 --steps 10 --warmup 2 --no-cpu-baseline > $O/t10_b5_f32.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/t10_b5_f32.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("f32:", d.get("ms_per_step"))
PY
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5e -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t10_p5.log 2>&1
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/stats5e/b_kernel_stats.csv")))[:8]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"])
PY
exit 0
