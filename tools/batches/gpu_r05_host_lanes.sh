#!/bin/bash
# GPU box (round 5): lanes of a delivered link-bound video once more (SDMA hand-off, transfer ramp up to 16+): KBE_HOST_LANES 2 (the default) / 3 / 4
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', d['config']['pass_ms'], 'lanes', d['config']['lanes'], d['frames_check']['ok'])"; }
for steps in 20 75 256; do
for l in 2 3 4 2; do
  echo "== --steps $steps KBE_HOST_LANES=$l"; KBE_HOST_LANES=$l timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | line
done
done
