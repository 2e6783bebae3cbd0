#!/bin/bash
# GPU box (round 4): timeline of a 20-frame delivered video (kernels per stream + the device-to-host copies), both ramps
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for ramp in fast classic; do
  rm -rf /tmp/o
  REPS=4 KBE_RAMP=$ramp HOST=1 FRAMES=${FRAMES:-20} timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/o -o t --output-format csv -- python $R/tools/frame_once.py > /tmp/o.log 2>&1 || tail -5 /tmp/o.log
  echo "== ramp $ramp"
  python $R/tools/handoff_timeline.py /tmp/o/t_kernel_trace.csv /tmp/o/t_memory_copy_trace.csv | tee $O/e_timeline_$ramp.txt
done
