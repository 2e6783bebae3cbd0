#!/bin/bash
# GPU box: one rocprofv3 --pmc pass per argument (a quoted counter list) over 9 frames; prints per-kernel means.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  rm -rf /tmp/p$i
  FRAMES=9 timeout 600 rocprofv3 --pmc $set -d /tmp/p$i -o c --output-format csv -- python $R/tools/frame_once.py > /tmp/p$i.log 2>&1 || tail -5 /tmp/p$i.log
  python $R/tools/pmc_kernels.py /tmp/p$i/c_counter_collection.csv
  i=$((i+1))
done
