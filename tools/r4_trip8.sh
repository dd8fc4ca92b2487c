#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/t8_coo.log 2>&1; rc=$?; tail -n 3 $O/t8_coo.log; [ $rc -ne 0 ] && exit $rc
for lr in 0 1; do
 if [ $lr = 1 ]; then export SPAL_COO_LOOP_RANKS=1; else unset SPAL_COO_LOOP_RANKS; fi
 SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/stamps/libspal_hip.so timeout -k 10 200 python bench.py --config 5 --steps 4 --warmup 1 --no-cpu-baseline > $O/t8_stamps_$lr.log 2>&1
 echo "LOOP_RANKS=$lr"; grep "spal coo stamps" $O/t8_stamps_$lr.log | tail -n 1
 timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/t8_b5_$lr.log 2>&1
 python - <<PY
import json
l=[x for x in open("$O/t8_b5_$lr.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("loop ranks $lr:", d.get("ms_per_step"), d.get("product_plan_ms"))
PY
done
exit 0
