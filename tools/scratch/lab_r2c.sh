#!/bin/bash
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -ci "utcl\|tlb" gpurun_out/counters_list.txt
timeout -k 10 300 python tools/lab_place.py pmc > gpurun_out/place_plain.log 2>&1; rc=$?; echo "plain rc=$rc"; cat gpurun_out/place_plain.log
[ $rc -ge 124 ] && exit $rc
for set in "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout -k 10 400 rocprofv3 --pmc $set --output-format csv -d gpurun_out/pmc_place_$tag -o p -- python3 tools/lab_place.py pmc > gpurun_out/pmc_place_$tag.log 2>&1; rc=$?
  echo "== $set rc=$rc"; tail -n 9 gpurun_out/pmc_place_$tag.log
  [ $rc -ge 124 ] && exit $rc
  f=$(find gpurun_out/pmc_place_$tag -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python tools/pmc_by_handle.py "$f"
done
