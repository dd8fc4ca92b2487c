#!/bin/bash
set -u
mkdir -p gpurun_out
for c in 1 2 3 4 5; do
  s=$(date +%s)
  timeout -k 10 600 python bench.py --config $c > gpurun_out/bench_cfg$c.log 2>&1
  rc=$?
  python3 - $c $rc $(( $(date +%s) - s )) <<'PY'
import json,sys
c,rc,secs=sys.argv[1:4]
for l in reversed(open(f'gpurun_out/bench_cfg{c}.log').read().splitlines()):
    if l.startswith('{'):
        j=json.loads(l); r=j.get('roofline',{}); cb=j.get('cpu_baseline',{})
        print(f"config {c} rc={rc} {secs}s: {j['value']:.1f} {j['unit']}  {j['ms_per_step']*1e3:.1f} us  frac {r.get('frac')}  cpu {cb.get('value')} {cb.get('unit')}  agrees {cb.get('gpu_agrees_with_cpu', cb.get('gpu_equals_cpu_bit_for_bit'))} {j.get('gpu_assembly_equals_cpu_bit_for_bit','')}")
        break
else:
    print(f"config {c} rc={rc}: no JSON"); print(open(f'gpurun_out/bench_cfg{c}.log').read()[-600:])
PY
  [ $rc -ge 124 ] && exit 1
done
exit 0
