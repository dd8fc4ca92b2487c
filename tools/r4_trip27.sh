#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/slstamps/libspal_hip.so timeout -k 10 300 python tools/lab.py shard "slide_on=1,nt_store=0" @rounds=2 > $O/t27_stamps.log 2>&1
grep "slide stamps" $O/t27_stamps.log | tail -n 4
grep median $O/t27_stamps.log | tail -n 1
exit 0
