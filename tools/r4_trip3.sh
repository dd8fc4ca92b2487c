#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/t3_coo.log 2>&1; rc=$?; tail -n 3 $O/t3_coo.log; [ $rc -ne 0 ] && exit $rc
for tk in 8 1 0; do
    SPAL_COO_TICKET=$tk timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t3_b5_$tk.log 2>&1; rc=$?
    python - <<PY
import json
l=[x for x in open("$O/t3_b5_$tk.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("ticket mode $tk", d.get("ms_per_step"), (d.get("roofline") or {}).get("route"))
PY
    [ $rc -ne 0 ] && exit $rc
done
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5c -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t3_p5.log 2>&1; rc=$?; echo "prof rc=$rc"
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/stats5c/b_kernel_stats.csv")))[:22]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"])
PY
