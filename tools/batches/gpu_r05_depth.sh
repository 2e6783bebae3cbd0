#!/bin/bash
# GPU box (round 5): the splat's look-ahead (KBE_SPLAT_DEPTH: steps of a wave whose points are in flight at once) on CONSECUTIVE cameras -- shared lists are longer
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_depth
mkdir -p $O
cd $R
for v in ${VARIANTS:-d4 d5 d3 d6 d4 d5 d3 d6}; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=12 PATHS=75,1024 LAUNCH_FRAMES=12 REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | tee -a $O/ahead_$v.txt | grep -E "consecutive|^12 frame|max \|diff\| [2-9]"
done
