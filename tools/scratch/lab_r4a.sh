#!/bin/bash
set -u
mkdir -p gpurun_out
for w in 16384 65536; do
  timeout -k 10 500 tools/micro/cblock_proto 10000000 $w 14 2>&1 | tee -a gpurun_out/cblock_proto.log
done
