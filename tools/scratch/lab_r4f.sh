#!/bin/bash
SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/coo_phases/libspal_hip.so timeout -k 10 300 python tools/lab_coo_once.py 2>&1 | grep -v amdgpu | tail -4
