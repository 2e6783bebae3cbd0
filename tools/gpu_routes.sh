#!/bin/bash
# GPU box: us per frame of the bench workload for the scatter routes / frames per launch / lanes (tools/throughput.py), frames
# left in HBM and (HOST=1) delivered (dev aid).   gpurun -- 'bash tools/gpu_routes.sh'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
for host in 0 1; do
for cfg in "KBE_FUSED=0 KBE_FILL_GROUP=1" "KBE_FUSED=0 KBE_FILL_GROUP=4" "KBE_FUSED=1 KBE_FILL_GROUP=1" "KBE_FUSED=1 KBE_FILL_GROUP=2" "KBE_FUSED=1 KBE_FILL_GROUP=4" "KBE_FUSED=1 KBE_FILL_GROUP=4 KBE_LANES=2" "KBE_FUSED=1 KBE_FILL_GROUP=4 KBE_LANES=3"; do
  echo "== HOST=$host $cfg $EXTRA_ENV: $(env HOST=$host $cfg $EXTRA_ENV FRAMES=${FRAMES:-512} REPS=5 timeout 300 python $R/tools/throughput.py 2>/dev/null | tail -1)"
done
done
