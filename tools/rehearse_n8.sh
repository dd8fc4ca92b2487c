#!/bin/bash
# dress rehearsal of the multi-GPU hosts at BASELINE config 3's FULL size on a ONE-GPU box.
# NOT a scaling measurement (every shard shares one GPU): what it shows is that the N-way partition, the x windows /
# broadcast, the local kernels, the gather / all-gather run end to end at their real sizes and agree with the oracle.
#  1. --host mg: EIGHT shards in one process through spal_mg_* (copy transport: RCCL has no ranks to talk to on one GPU);
#  2. the driver's launch form (torch.distributed.run, one process per rank) with FIVE ranks sharing the GPU over gloo -- the
#     pool allows six processes on a card and the launcher is one of them, so N = 8 of this form cannot run here.
set -u
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 600 python bench.py --host mg --gpus 8 --devices 0,0,0,0,0,0,0,0 --steps 20 --warmup 5 > $O/rehearsal_mg_8_shards.log 2>&1; rc=$?
grep "^{" $O/rehearsal_mg_8_shards.log | tail -n 1 > $O/rehearsal_mg_8_shards.json; echo "mg host rc=$rc $(wc -c < $O/rehearsal_mg_8_shards.json) bytes"
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 5 --master-addr 127.0.0.1 --master-port 29766 bench.py --gpus 5 --steps 20 --warmup 5 --backend gloo --same-device > $O/rehearsal_dist_5_ranks.log 2>&1; rc=$?
grep "^{" $O/rehearsal_dist_5_ranks.log | tail -n 1 > $O/rehearsal_dist_5_ranks.json; echo "dist host rc=$rc $(wc -c < $O/rehearsal_dist_5_ranks.json) bytes"
exit $rc
