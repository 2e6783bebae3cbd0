#!/bin/bash
# GPU box (round 5): why does a frame at the ends of the path cost 7 % more than one at its middle?  Twelve copies of one camera per launch at
# positions 0 / 0.5 / 1 of the path under the counters (one process per position and counter set)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for pos in 0 0.5 1; do
  for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"; do
    tag=$(echo $c | cut -d' ' -f1)
    rm -rf /tmp/pp
    IDENTICAL=0 PATHS= SKIP_CHECK=1 POSITIONS=$pos REPS=10 timeout 300 rocprofv3 --pmc $c -d /tmp/pp -o c --output-format csv -- python $R/tools/ahead_time.py > /tmp/pp.log 2>&1 || tail -2 /tmp/pp.log
    echo "position $pos: $(python $R/tools/pmc_by_grid.py /tmp/pp/c_counter_collection.csv k_frame_group_ahead | grep 'grid=6291456' | cut -c40-260)"
  done
done
