#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  for env in "SIZE=1024 UPSAMPLE=3 CLOUD=raw" "SIZE=512 UPSAMPLE=3 CLOUD=raw" "SIZE=512 UPSAMPLE=4 CLOUD=raw" "SIZE=1024"; do
    echo "== $v $env KBE_FUSED=1: $(KBE_LIB_PATH=$R/_variants/$v env $env KBE_FUSED=1 FRAMES=32 REPS=2 timeout 300 python $R/tools/throughput.py 2>&1 | tail -1)"
  done
done 2>&1 | tee $O/m_cliff.txt
