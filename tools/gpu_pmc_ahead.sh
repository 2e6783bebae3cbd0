#!/bin/bash
# GPU box: PMC passes over tools/ahead_time.py (the scatter launches alone on a stream), per kernel and frames per launch (dev aid).
#   gpurun -- 'bash tools/gpu_pmc_ahead.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" ...'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  rm -rf /tmp/pa$i
  REPS=8 timeout 600 rocprofv3 --pmc $set -d /tmp/pa$i -o c --output-format csv -- python $R/tools/ahead_time.py > /tmp/pa$i.log 2>&1 || tail -5 /tmp/pa$i.log
  python $R/tools/pmc_by_grid.py /tmp/pa$i/c_counter_collection.csv k_frame_group_ahead k_frame_group k_place
  i=$((i+1))
done
