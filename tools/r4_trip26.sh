#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py tests/test_gpu_bench.py -x -q -k "csc or config4" > $O/t26_tests.log 2>&1; rc=$?; tail -n 3 $O/t26_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/t26_tests.log | head; exit $rc; }
for rep in 1 2 3; do
for v in default rtprev; do
  unset SPAL_HIP_LIB; [ $v != default ] && export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so
  timeout -k 10 200 python bench.py --config 4 --steps 200 --warmup 20 --no-cpu-baseline > $O/t26_b4_$v.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/t26_b4_$v.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
print("$v:", d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), d.get("config",{}).get("plan",{}).get("row_tile_rows"), d.get("config",{}).get("plan",{}).get("row_tile_count"))
PY
done
done
exit 0
