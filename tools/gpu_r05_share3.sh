#!/bin/bash
# GPU box (round 5): round 4's tree against this one on identical cameras (did the sub-group sharing lose what the shared lists gained?)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2; do
echo "== r4 tree"; (cd $R/_variants/r4_tree && REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | grep -E "^(12| 8) frame") 
echo "== this tree"; IDENTICAL=12 PATHS= REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | grep -E "^12 frame"
echo "== this tree, px0"; KBE_LIB_PATH=$R/_variants/px0.so IDENTICAL=12 PATHS= REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | grep -E "^12 frame"
done
