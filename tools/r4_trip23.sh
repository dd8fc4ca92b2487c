#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_pl -o b -- python3 tools/lab.py powerlaw quick local > $O/t23_pl.log 2>&1
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/stats_pl/b_kernel_stats.csv")))[:10]:
    print(r["Name"][:90].ljust(90), r["Calls"], r["AverageNs"])
PY
exit 0
