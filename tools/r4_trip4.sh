#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py tests/test_gpu_cblock.py -x -q > $O/t4_tests.log 2>&1; rc=$?; tail -n 5 $O/t4_tests.log; [ $rc -ne 0 ] && exit $rc
for tk in 8 0; do
    SPAL_COO_TICKET=$tk timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t4_b5_$tk.log 2>&1; rc=$?
    python - <<PY
import json
l=[x for x in open("$O/t4_b5_$tk.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("ticket mode $tk", d.get("ms_per_step"), d.get("product_plan_ms"), d.get("spmv_on_result"))
PY
    [ $rc -ne 0 ] && exit $rc
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5d -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t4_p5.log 2>&1; rc=$?; echo "prof rc=$rc"
python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats5d/b_kernel_trace.csv")))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'coo_group_sort' in r['Kernel_Name']]
start=idx[-3]+1; last=idx[-2]
t0=int(rows[start]['Start_Timestamp']); prev=None
for r in rows[start:last+4]:
    s=int(r['Start_Timestamp']); e=int(r['End_Timestamp'])
    print(f"{(s-t0)/1e3:9.1f} us dur {(e-s)/1e3:8.1f} gap {((s-prev)/1e3 if prev else 0):7.1f} {r['Kernel_Name'][:60]}")
    prev=e
PY
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/t4_all.log 2>&1; rc=$?; tail -n 5 $O/t4_all.log; exit $rc
