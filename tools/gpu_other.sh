#!/bin/bash
# GPU box: the other workloads (512^2, raw cloud, dolly, 2048^2 raw, 2048^2 from 16.8 M points), both scatter routes, frames left in
# HBM (tools/throughput.py; HOST=1: delivered) (dev aid).
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
for env in "SIZE=256" "SIZE=512" "CLOUD=raw" "DOLLY=1" "SIZE=2048 CLOUD=raw" "SIZE=2048 UPSAMPLE=2 CLOUD=raw"; do
  for fused in 0 1; do
    echo "== $env KBE_FUSED=$fused $EXTRA_ENV: $(env $env KBE_FUSED=$fused $EXTRA_ENV FRAMES=128 REPS=3 timeout 300 python $R/tools/throughput.py 2>&1 | tail -1)"
  done
done
