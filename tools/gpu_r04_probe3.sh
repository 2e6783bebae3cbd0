#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/tools/short_video_probe.py 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tee $O/d_short_probe.txt | head -90
bash $R/tools/gpu_r04_variants.sh "$@" 2>&1 | grep -v "amdgpu.ids\|MIOpen" | grep "variant\|^ 8 \|^12 " | tee $O/d_variants.txt
