#!/bin/bash
# lab: COO tests, then config 5 three times and its per-kernel averages (rocprofv3 --kernel-trace --stats)
set -u
export TMPDIR=/tmp
O=$PWD/gpurun_out/r4; mkdir -p $O
[ -z "${SKIP_TESTS:-}" ] && { timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/cs_tests.log 2>&1; rc=$?; tail -n 2 $O/cs_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/cs_tests.log | head; exit $rc; }; }
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/cs_b5.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/cs_b5.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("ms per assembly:", d.get("ms_per_step"), "bit-exact:", d.get("config",{}).get("gpu_assembly_equals_cpu_bit_for_bit"))
PY
done
rm -rf $O/cs_prof; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cs_prof -o cs -- python3 $GRAFT_REPO_ROOT/bench.py --config 5 --steps 6 --warmup 1 --no-cpu-baseline > $O/cs_prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find $O/cs_prof -name "*kernel_stats.csv" | head -1)
python - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows[:12]:
    print("%-60s calls %5s avg %9.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3))
PY
