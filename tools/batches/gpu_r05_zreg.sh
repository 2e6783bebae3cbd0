#!/bin/bash
# GPU box (round 5): the degridded z in registers instead of LDS, and the records the freed 2 KB hold (LEAN_CAP 608 .. 680)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_zreg
mkdir -p $O
cd $R
for v in ${VARIANTS:-base z608 z640 z656 z672 z680 base z608 z640 z656 z672 z680}; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=12 PATHS=75 LAUNCH_FRAMES=12 REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | tee -a $O/ahead_$v.txt | grep -E "consecutive|^12 frame|max \|diff\| [2-9]"
done
