#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python tools/lab.py zoo 2>&1 | grep -v amdgpu.ids | tee $O/t21_zoo.txt | tail -n 12
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pl -o b -- python3 tools/lab.py powerlaw quick local > $O/t21_pl.log 2>&1
tail -n 8 $O/t21_pl.log
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/stats_pl/b_kernel_stats.csv")))[:8]:
    print(r["Name"][:70].ljust(70), r["Calls"], r["AverageNs"])
PY
exit 0
