#!/bin/bash
set -u
timeout -k 10 300 tools/micro/stream_ceiling --walk 2>&1 | head -3
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/diag/libspal_hip.so timeout -k 10 500 python tools/lab_ab1.py "diag=0,slide_on=1" "diag=2048,slide_on=1" "diag=6144,slide_on=1" "diag=0,slide_on=0" @rounds=3 2>&1 | grep -v amdgpu
timeout -k 10 300 tools/micro/stream_ceiling --walk 2>&1 | head -3
