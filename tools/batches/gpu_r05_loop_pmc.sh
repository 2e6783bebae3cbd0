#!/bin/bash
# GPU box (round 5): where a frame's vector instructions go in the frame loop of the DEFAULT workload, per kernel (one PMC pass over a 128-frame video left in HBM)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pdl
FRAMES=128 REPS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES -d /tmp/pdl -o c --output-format csv -- python $R/tools/throughput.py > /tmp/pdl.log 2>&1 || tail -3 /tmp/pdl.log
python $R/tools/pmc_by_grid.py /tmp/pdl/c_counter_collection.csv 2>&1 | cut -c1-260
tail -2 /tmp/pdl.log
