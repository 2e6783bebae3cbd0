#!/bin/bash
# GPU box (round 5): shared lists, second look -- identical cameras and real paths on the same builds, alternating
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_share2
mkdir -p $O
cd $R
for v in px16 px0 px16 px0; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=12 PATHS=1024,75 LAUNCH_FRAMES=12 POSITIONS=0,0.5,1 REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | tee -a $O/ahead_$v.txt | grep -E "consecutive|per launch:|camera at|max \|diff\| [2-9]"
done
