#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/final_gpu_tests.log 2>&1; rc=$?; tail -n 4 $O/final_gpu_tests.log; [ $rc -ne 0 ] && { grep -n "Error\|assert\|error" $O/final_gpu_tests.log | head -20; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" > $O/final_smoke.log 2>&1; echo "smoke rc=$?"; tail -n 1 $O/final_smoke.log | cut -c1-300
( time timeout -k 10 600 python bench.py ) > $O/final_bench.log 2>&1; echo "bench rc=$?"
grep "^{" $O/final_bench.log | tail -n 1 > $O/final_bench.json
python - <<PY
import json
d=json.load(open("$O/final_bench.json"))
r=d["roofline"]
print("headline", d["value"], d["ms_per_step"], "frac", r["frac"], "moved_frac", r.get("moved_frac"), "hbm_frac", r.get("hbm_frac"), "quote", r.get("frac_to_quote"))
for k,v in d["other_configs"].items():
    if "error" in v: print(k, "ERROR", v); continue
    rr=v["roofline"]
    print(k, v["ms_per_step"], "frac", rr.get("frac"), "hbm_frac", rr.get("hbm_frac"), rr.get("frac_to_quote"), "parity", v.get("parity"), {q: v.get(q) for q in ("product_plan_ms","ms_per_step_plus_product_plan")}, (v.get("spmv_on_result") or {}).get("ms"))
PY
grep real $O/final_bench.log
