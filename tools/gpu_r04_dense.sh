#!/bin/bash
# GPU box (round 4): clouds denser than the raster, both routes (tree build), frames left in HBM and delivered; then the at-size tests
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R && timeout 900 python -m pytest tests/test_hip_at_size.py -x -q -m gpu 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for env in "SIZE=2048 UPSAMPLE=2 CLOUD=raw" "SIZE=1024 UPSAMPLE=2 CLOUD=raw" "SIZE=512 UPSAMPLE=2 CLOUD=raw" "SIZE=1024 UPSAMPLE=3 CLOUD=raw"; do
  for fused in 1 0; do
    for host in 0 1; do
      echo "== $env KBE_FUSED=$fused HOST=$host: $(env $env KBE_FUSED=$fused HOST=$host FRAMES=64 REPS=3 timeout 300 python $R/tools/throughput.py 2>&1 | tail -1)"
    done
  done
done 2>&1 | tee $O/j_dense_routes.txt
