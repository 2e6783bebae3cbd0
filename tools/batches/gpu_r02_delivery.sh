#!/bin/bash
# GPU box: all GPU tests, the default bench, a few hand-off settings (dev aid, round 2)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${OUT:-r02i}
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/pytest.log
for hl in 2 3; do for b in -8 -16; do
    echo "== host lanes $hl batch $b" >> $O/sweep.log
    BATCH=$b KBE_HOST_LANES=$hl HOST=1 FRAMES=1024 REPS=4 timeout 300 python tools/throughput.py 2>/dev/null | tail -1 >> $O/sweep.log
done; done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/pytest.log $O/sweep.log; head -c 2500 $O/bench.json
