#!/bin/bash
# GPU box (round 4): short delivered videos with the transfers taking no turns, by lanes
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.0f frames/s (%.3f ms per pass, lanes %s, ok %s)' % (d['value'], d['config']['pass_ms']['median'], d['config']['lanes'], d['frames_check']['ok']))"; }
for steps in 20 75; do
for env in "KBE_FREE_TRANSFERS=0" "KBE_FREE_TRANSFERS=1" "KBE_FREE_TRANSFERS=1 KBE_HOST_LANES=3" "KBE_FREE_TRANSFERS=1 KBE_HOST_LANES=4" "KBE_FREE_TRANSFERS=1 KBE_RAMP=fast" "KBE_HOST_LANES=1"; do
  echo "steps $steps [$env]: $(env $env timeout 300 python bench.py --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | val)" | tee -a $O/groups.txt
done; done
