#!/bin/bash
# GPU box: frames left in HBM (or HOST=1: delivered), us per frame by lanes x frames per launch of the fused route (dev aid).
#   gpurun -- 'bash tools/gpu_shape_sweep.sh "1 2 3 4" "1 2 4 8 12"'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lanes in ${1:-1 2 4}; do
  for group in ${2:-1 2 4 8 12}; do
    echo -n "lanes $lanes frames/launch $group: "
    KBE_LANES=$lanes KBE_HOST_LANES=$lanes KBE_FILL_GROUP=$group FRAMES=${FRAMES:-480} REPS=${REPS:-4} timeout 300 python $R/tools/throughput.py 2>/dev/null | tail -1
  done
done
