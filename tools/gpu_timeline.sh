#!/bin/bash
# GPU box: kernel trace of the host-delivered frame loop, per lane count (dev aid)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lanes in ${LANES:-1 4}; do
  rm -rf /tmp/o
  KBE_LANES=$lanes FRAMES=${FRAMES:-129} timeout 600 rocprofv3 --kernel-trace -d /tmp/o -o t --output-format csv -- python $R/tools/frame_once.py > /dev/null 2>&1
  echo "== lanes $lanes"
  python $R/tools/timeline.py /tmp/o/t_kernel_trace.csv 16
  mkdir -p $R/gpurun_out/${OUT:-r02d}; cp /tmp/o/t_kernel_trace.csv $R/gpurun_out/${OUT:-r02d}/trace_lanes$lanes.csv
done
