#!/bin/bash
# lab: the block-window kernel -- its tests, the power-law lab matrix (columns near the rows), the phase stamps of a lab build
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_csr_spmv.py tests/test_gpu_csr_fuzz.py -x -q -k "block_window or random_matrices" > $O/bw_tests.log 2>&1; rc=$?; tail -n 2 $O/bw_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert" $O/bw_tests.log | head; exit $rc; }
timeout -k 10 300 python tools/lab.py powerlaw quick > $O/pl_bw.log 2>&1; grep "blockwin\|power-law" $O/pl_bw.log | head -12
if [ -f spalinalg_amd/lib_var/bwstamps/libspal_hip.so ]; then
  SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/bwstamps/libspal_hip.so timeout -k 10 300 python tools/lab.py powerlaw local quick > $O/pl_bws.log 2>&1; grep "blockwin stamps" $O/pl_bws.log | head -2
fi
for v in ${VARIANTS:-}; do SPAL_HIP_LIB=$PWD/spalinalg_amd/lib_var/$v/libspal_hip.so timeout -k 10 300 python tools/lab.py powerlaw local quick > $O/pl_bw_$v.log 2>&1; echo "$v: $(grep blockwin $O/pl_bw_$v.log | head -1)"; done
exit 0
