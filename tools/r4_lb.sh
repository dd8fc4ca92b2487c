#!/bin/bash
# lab: look-back variants of coo_group_sort side by side (libraries under spalinalg_amd/lib_var/<name>), config 5
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
VARIANTS=${VARIANTS:-"default lbw2 lbw4 lbw8"}
TESTLIB=${TESTLIB:-}
[ -n "$TESTLIB" ] && export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$TESTLIB/libspal_hip.so
timeout -k 10 600 python -m pytest tests/test_gpu_csc_coo.py -x -q -k "coo" > $O/lb_tests.log 2>&1; rc=$?; tail -n 3 $O/lb_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/lb_tests.log | head; exit $rc; }
for rep in 1 2 3; do
for v in $VARIANTS; do
  # <library>[_loop]: the _loop suffix runs the library with SPAL_COO_LOOP_RANKS=1 (the group kernel's other form)
  unset SPAL_HIP_LIB SPAL_COO_LOOP_RANKS; lib=${v%_loop}; [ $lib != $v ] && export SPAL_COO_LOOP_RANKS=1
  [ $lib != default ] && export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$lib/libspal_hip.so
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
