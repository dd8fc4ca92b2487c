#!/bin/bash
# HBM bytes of the block-window kernel on the power-law lab matrix: rocprofv3 --pmc, one counter per pass, nothing else beside it
# (FETCH_SIZE in KiB x 2 on gfx950, WRITE_SIZE in KiB: profiles/r04/pmc_traffic.txt)
set -u
export TMPDIR=/tmp
D=$PWD/gpurun_out/r4/pmc_bw; mkdir -p $D
O=$PWD/gpurun_out/r4/pmc_blockwin.txt
echo "# rocprofv3 --pmc <counter> -- python3 tools/lab.py powerlaw local quick ; kernel csr_spmv_blockwin; mean per launch" > $O
cd /tmp
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES; do
  rm -rf $D/$c
  timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d $D/$c -o b -- python3 $GRAFT_REPO_ROOT/tools/lab.py powerlaw local quick > $D/$c.log 2>&1 || { echo "$c failed" >> $O; continue; }
  python3 - $c $D >> $O <<'P'
import csv,sys
c,d=sys.argv[1:3]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f"{d}/{c}/b_counter_collection.csv")) if "blockwin" in r["Kernel_Name"]]
scale = 2048.0 if c == "FETCH_SIZE" else 1024.0 if c == "WRITE_SIZE" else 1.0
unit = " bytes" if c in ("FETCH_SIZE", "WRITE_SIZE") else ""
print(f"{c:24s} launches {len(v):3d}  mean {sum(v)/max(1,len(v))*scale:.6g}{unit}")
P
done
cat $O
