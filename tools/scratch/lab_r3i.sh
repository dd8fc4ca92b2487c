#!/bin/bash
set -u
export TMPDIR=/tmp
P=gpurun_out/prof
mkdir -p $P; rm -rf $P/tr_ovb
timeout -k 10 420 rocprofv3 --kernel-trace --output-format csv -d $P/tr_ovb -o b -- python3 tools/lab_ab1.py "overflow_beside=1" ragged @rounds=1 > $P/tr_ovb.log 2>&1
echo "rc=$?"
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/prof/tr_ovb/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'csr_spmv' in r['Kernel_Name']]
rows=rows[-12:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    print(f"{r['Kernel_Name'][12:40]:30s} q={r.get('Queue_Id','?')} start {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} us  end {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us")
PY
