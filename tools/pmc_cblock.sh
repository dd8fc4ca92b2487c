# counters of the column-blocked kernels (rocprofv3 --pmc, one counter per pass): `bash tools/pmc_cblock.sh [nrows per_row]`
# (default: the 5M x 5M, 10-per-row matrix); COUNTERS="..." selects
set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3
TAG=${1:-5M}
O=gpurun_out/r3/pmc_cblock_$TAG.txt
: > $O
for c in ${COUNTERS:-TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE}; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d gpurun_out/r3/pmcb_$c -o b -- python3 tools/lab.py cblock_once "$@" > gpurun_out/r3/pmcb_$c.log 2>&1 || { echo "$c failed" >> $O; continue; }
  python3 - $c >> $O <<'P'
import csv,sys
c=sys.argv[1]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f"gpurun_out/r3/pmcb_{c}/b_counter_collection.csv")) if "csr_spmv_cblock" in r["Kernel_Name"]]
print(f"{c:32s} launches {len(v):3d}  mean {sum(v)/max(1,len(v)):.4g}")
P
done
cat $O
