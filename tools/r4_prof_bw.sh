#!/bin/bash
# per-kernel stats of the power-law lab matrix (block-window kernel against the row split): rocprofv3 --kernel-trace --stats
set -u
export TMPDIR=/tmp
P=$PWD/gpurun_out/prof4; mkdir -p $P
rm -rf $P/statspl
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $P/statspl -o b -- python3 $GRAFT_REPO_ROOT/tools/lab.py powerlaw local quick > $P/statspl.log 2>&1
echo "rc=$?"; grep -v "Warn\|amdgpu.ids" $P/statspl.log | grep "blockwin\|split" | head -4
cd $GRAFT_REPO_ROOT
python - <<PY
import csv,glob
f=glob.glob("$P/statspl/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:8]:
    print("%-70s calls %5s avg %9.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
