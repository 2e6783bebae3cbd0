#!/bin/bash
# GPU box: one rocprofv3 --pmc pass per argument (a quoted counter list) over tools/scatter_time.py (both routes' scatter
# launches with 1..4 frames each); prints per (kernel, frames per launch) means.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  rm -rf /tmp/q$i
  GROUP_ONLY=4 timeout 600 rocprofv3 --pmc $set -d /tmp/q$i -o c --output-format csv -- python $R/tools/scatter_time.py > /tmp/q$i.log 2>&1 || tail -5 /tmp/q$i.log
  python $R/tools/pmc_by_grid.py /tmp/q$i/c_counter_collection.csv k_place k_frame k_project k_tiles | grep -E "x1|x4"
  i=$((i+1))
done
