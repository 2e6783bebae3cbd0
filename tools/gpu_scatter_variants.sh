#!/bin/bash
# GPU box: for each variant (a quoted set of extra hipcc flags) build the library into /tmp and time the scatter of both
# routes alone on a stream with 1..4 frames per launch (tools/scatter_time.py) (dev aid).
#   gpurun -- 'bash tools/gpu_scatter_variants.sh "" "-DKBE_TILE_H=32 -DKBE_TILE_THREADS=512 -DKBE_TILE_CAP=1536"'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
i=0
for flags in "$@"; do
  so=/tmp/libkbe_var_$i.so
  make -s -B -C $R/ken-burns-effect_amd/csrc EXTRA="$flags" OUT=$so || exit 1
  echo "== variant: ${flags:-(default)}"
  KBE_LIB_PATH=$so timeout 600 python $R/tools/scatter_time.py 2>&1 | grep -E "per frame|fused:scatter" | cut -c1-200
  i=$((i+1))
done
