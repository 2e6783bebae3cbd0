#!/bin/bash
# GPU box (round 4): the scatter alone on a stream (tools/ahead_time.py, 8 and 12 frames per launch shown) for prebuilt libraries
# (_variants/NAME.so, built here beforehand: e.g. the round's starting point) and for variant builds (quoted extra hipcc flags).
#   gpurun -- 'bash tools/batches/gpu_r04_variants.sh head.so "" "-DKBE_FRAME_WAVES=6 -DKBE_TILE_CAP=704"'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for v in "$@"; do
  if [ -f "$R/_variants/$v" ]; then so=$R/_variants/$v; else
    so=/tmp/libkbe_var_$i.so
    make -s -B -C $R/ken-burns-effect_amd/csrc EXTRA="$v -Rpass-analysis=kernel-resource-usage" OUT=$so 2>&1 | grep -A8 "19k_frame_group_aheadE" | grep -E "VGPRs:|Spill|LDS Size|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo
  fi
  echo "== variant: ${v:-(tree)}"
  KBE_LIB_PATH=$so REPS=${REPS:-40} timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -E "frame\(s\) per launch|max \|diff\| [2-9]" | tee -a $O/variants.txt
  i=$((i+1))
done
