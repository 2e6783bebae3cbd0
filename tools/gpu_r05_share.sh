#!/bin/bash
# GPU box (round 5): candidate lists shared in sub-groups on REAL paths -- the share threshold (KBE_SHARE_MAX_PX variants), tile counters,
# and the GPU suite on the shipped library
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_share
mkdir -p $O
cd $R
for v in px0 px4 px8 px16 px32 px1e9 px0 px16; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=0 PATHS=1024,75,20 GROUPS=12 REPS=40 timeout 600 python tools/ahead_time.py 2>&1 | tee -a $O/ahead_$v.txt | grep -E "consecutive|max \|diff\| [2-9]"
done
for v in stats stats_px1e9; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=0 PATHS=1024,75,20 GROUPS=12 REPS=12 timeout 600 python tools/ahead_time.py 2>&1 | tee $O/ahead_$v.txt | grep -E "consecutive"
done
echo "== gpu tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
