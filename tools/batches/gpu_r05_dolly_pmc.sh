#!/bin/bash
# GPU box (round 5): where a dolly frame's vector instructions go, per kernel (one PMC pass over a 64-frame dolly video, frames left in HBM)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pdl
DOLLY=1 FRAMES=64 REPS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES -d /tmp/pdl -o c --output-format csv -- python $R/tools/throughput.py > /tmp/pdl.log 2>&1 || tail -3 /tmp/pdl.log
python $R/tools/pmc_by_grid.py /tmp/pdl/c_counter_collection.csv 2>&1 | cut -c1-260
rm -rf /tmp/kdl
DOLLY=1 FRAMES=128 REPS=2 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kdl -o b --output-format csv -- python $R/tools/throughput.py > /tmp/kdl.log 2>&1
head -12 /tmp/kdl/b_kernel_stats.csv | cut -c1-200
tail -2 /tmp/kdl.log
