#!/bin/bash
# GPU box (round 4): tools/turn_check.py (four ranks on one GPU) $1 times; the output of the runs that fail
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $R/gpurun_out/turn
for i in $(seq 1 ${1:-12}); do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29600 + i)) $R/tools/turn_check.py > $R/gpurun_out/turn/$i.log 2>&1
  rc=$?
  echo "run $i: rc $rc $(grep -c 'OK (' $R/gpurun_out/turn/$i.log)"
  if [ $rc -ne 0 ]; then grep "AssertionError\|KbeError" $R/gpurun_out/turn/$i.log | head -4 | cut -c1-600; fi
done
