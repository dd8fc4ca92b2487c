#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/lb_tests.log 2>&1; rc=$?; tail -n 3 $O/lb_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/lb_tests.log | head; exit $rc; }
for rep in 1 2 3; do
for v in default nospec; do
  unset SPAL_HIP_LIB; [ $v != default ] && export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so
  timeout -k 10 200 python bench.py --config 5 --steps 10 --warmup 2 --no-cpu-baseline > $O/lb_b5_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/lb_b5_$v.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v:", d.get("ms_per_step"))
PY
done
done
exit 0
