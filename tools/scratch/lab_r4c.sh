#!/bin/bash
set -u
mkdir -p gpurun_out
for args in "--dtype f32" "--dist ragged" "--config 2" "--config 2 --dist uniform" "--dist uniform --steps 20 --warmup 3"; do
  timeout -k 10 400 python bench.py $args --no-cpu-baseline --no-ceiling > gpurun_out/b_misc.log 2>&1
  rc=$?
  python3 - "$args" <<'PY'
import json,sys
for l in reversed(open('gpurun_out/b_misc.log').read().splitlines()):
    if l.startswith('{'):
        j=json.loads(l); r=j['roofline']; p=j['config'].get('plan',{})
        print(f"{sys.argv[1]:45s} {j['ms_per_step']*1e3:8.1f} us  {j['value']:8.1f} {j['unit']}  frac {r['frac']}  kernel {r['kernel']}  rpt {p.get('rows_per_tile')} slide {p.get('slide')} lds_x {p.get('lds_x')}")
        break
else:
    print(sys.argv[1], "no json")
PY
  [ $rc -ge 124 ] && exit 1
done
