#!/bin/bash
# GPU box (round 5): the share threshold once more (10 / 13 / 16 / 20 px), alternating, on the 75- and 20-step paths; then the GPU suite
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_sixth
mkdir -p $O
cd $R
for rep in 1 2 3; do
for v in px10 px13 px16 px20; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=0 SKIP_CHECK=1 PATHS=75,20 LAUNCH_FRAMES=12 REPS=60 timeout 600 python tools/ahead_time.py 2>&1 | tee -a $O/ahead_$v.txt | grep -E "consecutive"
done
done
echo "== gpu tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
