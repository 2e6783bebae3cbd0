#!/bin/bash
# GPU box (round 5): the other workloads (frames left in HBM), both routes where it matters -- against profiles/r04_other_workloads.txt
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for env in "SIZE=512" "CLOUD=raw" "DOLLY=1" "SIZE=2048 CLOUD=raw" "SIZE=2048 UPSAMPLE=2 CLOUD=raw"; do
  for fused in 0 auto; do
    echo "== $env KBE_FUSED=$fused"
    env $env KBE_FUSED=$fused FRAMES=128 REPS=3 timeout 300 python $R/tools/throughput.py 2>/dev/null | tail -1
  done
done
