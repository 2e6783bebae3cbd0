#!/bin/bash
# GPU box: frames left in HBM, us per frame, for variant builds of the library; the workload from the environment (SIZE, UPSAMPLE,
# CLOUD, DOLLY, KBE_FUSED, FRAMES, ...).   gpurun -- 'SIZE=2048 UPSAMPLE=2 CLOUD=raw KBE_FUSED=1 bash tools/gpu_variant_throughput.sh "" "-DKBE_TILE_CAP=1472 -DKBE_FRAME_WAVES=3"'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  so=/tmp/libkbe_var_$i.so
  make -s -B -C $R/ken-burns-effect_amd/csrc EXTRA="$flags" OUT=$so || exit 1
  echo -n "variant '${flags}': "
  KBE_LIB_PATH=$so FRAMES=${FRAMES:-128} REPS=${REPS:-3} timeout 600 python $R/tools/throughput.py 2>/dev/null | tail -1
  i=$((i+1))
done
