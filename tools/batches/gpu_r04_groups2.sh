#!/bin/bash
# GPU box (round 4): delivered videos with a switch of the hand-off on and off (arguments: the environment settings to compare)
#   bash tools/batches/gpu_r04_groups2.sh "KBE_EARLY_TURN=0" "KBE_EARLY_TURN=1"
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.0f frames/s (%.3f ms per pass, lanes %s, ok %s)' % (d['value'], d['config']['pass_ms']['median'], d['config']['lanes'], d['frames_check']['ok']))"; }
for steps in 20 75 1024; do
for env in "$@" "$@"; do
  w=20; [ $steps = 1024 ] && w=128
  echo "steps $steps [$env]: $(env $env timeout 300 python bench.py --steps $steps --warmup $w --no-cpu-baseline 2>/dev/null | val)" | tee -a $O/handoff_switch.txt
done; done
