#!/bin/bash
# GPU box (round 4): short delivered videos by lanes and ramp
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
for lanes in 2 3 4; do
 for ramp in classic fast; do
  for steps in 20 75; do
    KBE_HOST_LANES=$lanes KBE_RAMP=$ramp timeout 600 python bench.py --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
    python - <<P
import json
d = json.loads(open('/tmp/line.json').read())
print('lanes $lanes ramp $ramp steps $steps: %.0f frames/s delivered (%.3f ms per pass), pcie %.1f GB/s, check %s' % (d['value'], d['ms_per_step'] * $steps, d['pcie']['achieved'], d['frames_check']['ok']))
P
  done
 done
done 2>&1 | tee $O/f_short_lanes.txt
