#!/bin/bash
# GPU box (round 5): the splat's winner corner from the fractions (near) against the four products (prod)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_winner
mkdir -p $O
cd $R
for v in ${VARIANTS:-prod near prod near prod near}; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=12 PATHS=75 LAUNCH_FRAMES=12 REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | tee -a $O/ahead_$v.txt | grep -E "consecutive|^12 frame|max \|diff\| [2-9]"
done
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
