#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/t2_coo.log 2>&1; rc=$?; tail -n 3 $O/t2_coo.log; [ $rc -ne 0 ] && exit $rc
for v in default wgs5; do
  if [ $v = default ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  for occ in 8 3; do
    SPAL_COO_GROUP_OCC=$occ timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t2_b5_${v}_$occ.log 2>&1; rc=$?
    python - <<PY
import json
l=[x for x in open("$O/t2_b5_${v}_$occ.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v occ<=$occ", d.get("ms_per_step"), d.get("gpu_assembly_equals_cpu_bit_for_bit"), (d.get("roofline") or {}).get("route",{}).get("group_grid"))
PY
    [ $rc -ne 0 ] && exit $rc
  done
done
unset SPAL_HIP_LIB
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats5b -o b -- python3 bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t2_p5.log 2>&1; rc=$?; echo "prof rc=$rc"
python - <<PY
import csv
for r in list(csv.DictReader(open("$O/stats5b/b_kernel_stats.csv")))[:14]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"])
PY
