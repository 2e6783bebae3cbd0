#!/bin/bash
# GPU box (round 5): does the candidate lists' capacity (512 / 768 / 1024 entries per tile) cost the headline launch anything?
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do
for v in cand512 cand768 cand1024; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=0 SKIP_CHECK=1 PATHS=75,1024 LAUNCH_FRAMES=12 REPS=60 timeout 600 python tools/ahead_time.py 2>&1 | grep -E "consecutive"
done
done
