#!/bin/bash
# lab: the headline twice (fresh processes), with what the autotune and the placement walk saw
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
for i in 1 2; do
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-other-configs > $O/head_$i.log 2>&1
  python - <<PY
import json
l=[x for x in open("$O/head_$i.log") if x.startswith("{")]
d=json.loads(l[-1]); p=d["config"]["plan"]
print(d["ms_per_step"], d["roofline"]["frac"], "slide", p.get("slide"), "autotune_us", p.get("autotune_us"), "walk_us", p.get("vectors_walk_us"), "placement_us", p.get("placement_us"))
PY
done
exit 0
