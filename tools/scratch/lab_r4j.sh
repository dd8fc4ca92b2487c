#!/bin/bash
for v in default static7 default static7; do
  if [ $v = default ]; then unset SPAL_HIP_LIB; else export SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so; fi
  echo "== $v"
  timeout -k 10 400 python tools/lab_ab1.py "slide_on=1" "slide_on=0" "slide_on=0,prefetch=2" @rounds=2 2>&1 | grep -v amdgpu | cut -c1-200
done
