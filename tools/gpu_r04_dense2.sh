#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for env in "SIZE=2048 UPSAMPLE=2 CLOUD=raw" "SIZE=1024 UPSAMPLE=2 CLOUD=raw"; do
  for knobs in "KBE_FUSED=1" "KBE_FUSED=1 KBE_FILL_GROUP=4" "KBE_FUSED=1 KBE_FILL_GROUP=4 KBE_HOST_LANES=4" "KBE_FUSED=1 KBE_FILL_GROUP=2 KBE_HOST_LANES=4" "KBE_FUSED=1 KBE_HOST_LANES=4" "KBE_FUSED=1 KBE_HOST_LANES=3 KBE_FILL_GROUP=4" "KBE_FUSED=0" "KBE_FUSED=0 KBE_HOST_LANES=2"; do
      echo "== $env $knobs HOST=1: $(env $env $knobs HOST=1 FRAMES=64 REPS=3 timeout 300 python $R/tools/throughput.py 2>&1 | tail -1)"
  done
done 2>&1 | tee $O/k_dense_delivered.txt
