#!/bin/bash
# GPU box (round 5): where the instructions of k_frame_group_ahead go on real cameras: -DKBE_FRAME_STOP=n builds (the kernel ends after stage n) under one PMC pass each
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in stop1 stop3 stop4 stop5 full; do
  rm -rf /tmp/ps_$v
  KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=0 PATHS=75 SKIP_CHECK=1 LAUNCH_FRAMES=12 REPS=8 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d /tmp/ps_$v -o c --output-format csv -- python $R/tools/ahead_time.py > /tmp/ps_$v.log 2>&1 || tail -3 /tmp/ps_$v.log
  echo "== $v: $(python $R/tools/pmc_by_grid.py /tmp/ps_$v/c_counter_collection.csv k_frame_group_ahead | grep 'grid=6291456' | cut -c1-200)"
done
