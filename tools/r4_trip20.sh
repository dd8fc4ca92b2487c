#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_csr_spmv.py -x -q > $O/t20_tests.log 2>&1; rc=$?; tail -n 3 $O/t20_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert" $O/t20_tests.log | head -20; exit $rc; }
for rep in 1 2; do
SPAL_WALK_DEBUG=1 timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-other-configs --cpu-seconds 2 > $O/t20_b3_$rep.log 2>&1
python - <<PY
import json
l=[x for x in open("$O/t20_b3_$rep.log") if x.startswith("{")]
d=json.loads(l[-1]) if l else {}
pl=d.get("config",{}).get("plan",{})
print("config 3:", d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"), d.get("setup_s"), {k: pl.get(k) for k in ("vectors_walk_us","vectors_walk_blocks","vectors_probes","placement_us","placement_tries","placement_blocks","placement_free_bytes")})
PY
grep "spal walk" $O/t20_b3_$rep.log | head -4
done
exit 0
