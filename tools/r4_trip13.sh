#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py tests/test_gpu_cblock.py -x -q > $O/t13_tests.log 2>&1; rc=$?; tail -n 3 $O/t13_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/t13_tests.log | head -20; exit $rc; }
for rep in 1 2; do
for v in default prev; do
  unset SPAL_HIP_LIB
  case $v in prev) export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/prev/libspal_hip.so;; esac
  timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t13_b5_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/t13_b5_$v.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v:", d.get("ms_per_step"), d.get("product_plan_ms"), (d.get("spmv_on_result") or {}).get("ms"))
PY
done
done
unset SPAL_HIP_LIB
for c in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_wide_$c -o w -- tools/micro/wide_scatter > $O/pmc_wide_$c.log 2>&1
done
python - <<PY
import csv, collections
for c in ("WRITE_SIZE","FETCH_SIZE"):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open("$O/pmc_wide_%s/w_counter_collection.csv" % c)):
        if "wide_scatter" in r["Kernel_Name"]: acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    for k,v in acc.items(): print(c, k[:60], "launches", len(v), "mean KiB", sum(v)/len(v))
PY
exit 0
