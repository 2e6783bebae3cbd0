#!/bin/bash
# GPU box (round 5): the device-side timeline of a 20-frame delivered video with the SDMA hand-off: kernel trace + memory-copy trace of a short run
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_timeline
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
FRAMES=${FRAMES:-20} PASSES=6 timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/tl -o t --output-format csv -- python $R/tools/short_pass.py > $O/probe.txt 2>&1
ls /tmp/tl; head -3 /tmp/tl/t_memory_copy_trace.csv
python $R/tools/sdma_timeline.py /tmp/tl/t_kernel_trace.csv /tmp/tl/t_memory_copy_trace.csv | tee $O/timeline.txt | tail -60
