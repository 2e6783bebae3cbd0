#!/bin/bash
# GPU box (round 5): shared lists in sub-groups, after the byte loads left the prologue: threshold variants on identical cameras and real paths
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_share4
mkdir -p $O
cd $R
echo "== r4 tree"; (cd $R/_variants/r4_tree && REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | grep -E "^12 frame")
for v in px16 px0 px4 px8 px32 px16 px0; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=12 PATHS=1024,75,20 LAUNCH_FRAMES=12 REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | tee -a $O/ahead_$v.txt | grep -E "consecutive|^12 frame|max \|diff\| [2-9]"
done
