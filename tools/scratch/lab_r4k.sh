#!/bin/bash
for i in 1 2; do
timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/b3_place.log 2>&1
python3 - <<'PY'
import json
for l in reversed(open('gpurun_out/b3_place.log').read().splitlines()):
    if l.startswith('{'):
        j=json.loads(l); p=j['config']['plan']; r=j['roofline']
        print(j['ms_per_step']*1e3, 'us', j['value'], 'frac', r['frac'], 'ceiling', r.get('ceiling',{}).get('footprint_us'), 'of ceiling', r.get('moved_frac_of_ceiling'), 'placement', p['placement_us'], p['placement_tries'], 'autotune', p['autotune_us'])
        break
PY
done
timeout -k 10 300 python -m pytest tests/test_gpu_csr_spmv.py -x -q -m gpu -k "autotune" 2>&1 | tail -2
