#!/bin/bash
# GPU box (round 5): the ramp of a short delivered video's transfer groups once more with the SDMA hand-off: classic 1, 2, 4, ... / fast 1, 3, 7, ... / even groups
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', d['config']['pass_ms'], d['frames_check']['ok'])"; }
for steps in 20 16 75; do
for spec in "" "KBE_RAMP=fast" "KBE_RAMP=fast KBE_DELIVERY_BATCH=-16" "KBE_EVEN_GROUPS=1 KBE_DELIVERY_BATCH=-4" "KBE_EVEN_GROUPS=1 KBE_DELIVERY_BATCH=-5" "KBE_EVEN_GROUPS=1 KBE_DELIVERY_BATCH=-7" "KBE_EVEN_GROUPS=1 KBE_DELIVERY_BATCH=-2" ""; do
  echo "== --steps $steps $spec"; env $spec timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | line
done
done
