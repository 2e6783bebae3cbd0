#!/bin/bash
# GPU box (round 4): tools/frame_probe.py for the variants given as arguments (each a quoted set of extra hipcc flags; "" = as shipped)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  echo "== variant: ${flags:-(as shipped)}"
  EXTRA="$flags" FRAMES=${FRAMES:-8} timeout 600 python $R/tools/frame_probe.py 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tee $O/frame_probe_v$i.txt | tail -22
  i=$((i+1))
done
