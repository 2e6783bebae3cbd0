#!/bin/bash
# GPU box: parity tests, then a clean kernel trace of 33 frames and the per-kernel medians (dev aid).
#   gpurun -- 'bash tools/gpu_ab.sh [notest]'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" != "notest" ]; then timeout 900 python -m pytest $R/tests -m gpu -x -q 2>&1 | tail -2; fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/o
KBE_LANES=1 FRAMES=${FRAMES:-33} timeout 600 rocprofv3 --kernel-trace -d /tmp/o -o t --output-format csv -- python $R/tools/frame_once.py > /dev/null 2>&1
python $R/tools/kernel_times.py /tmp/o/t_kernel_trace.csv
python $R/tools/throughput.py 2>/dev/null | tail -1
