#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "csc" > $O/t19_tests.log 2>&1; rc=$?; tail -n 3 $O/t19_tests.log; [ $rc -ne 0 ] && exit $rc
for rep in 1 2 3; do
for v in default rtprev; do
  unset SPAL_HIP_LIB; [ $v != default ] && export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so
  timeout -k 10 200 python bench.py --config 4 --steps 200 --warmup 20 --no-cpu-baseline > $O/t19_b4_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/t19_b4_$v.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v:", d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("transposed_route") or {}).get("ms_per_step"))
PY
done
done
unset SPAL_HIP_LIB
timeout -k 10 200 python bench.py --config 2 --steps 200 --warmup 20 --no-cpu-baseline --no-ceiling > $O/t19_b2.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/t19_b2.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("config 2:", d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), (d.get("roofline") or {}).get("kernel"))
PY
exit 0
