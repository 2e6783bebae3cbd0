#!/bin/bash
# GPU box (round 4): configs[4] (2048^2 from 16.8 M points), frames left in HBM, by library variant and route
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for fused in ${ROUTES:-1 0}; do
    echo "== $v KBE_FUSED=$fused: $(KBE_LIB_PATH=$R/_variants/$v SIZE=2048 UPSAMPLE=2 CLOUD=raw KBE_FUSED=$fused FRAMES=64 REPS=3 timeout 300 python $R/tools/throughput.py 2>&1 | tail -1)"
  done
done 2>&1 | tee $O/h_c4_variants.txt
