#!/bin/bash
# GPU box (round 5): the transfer-group cap of SHORT delivered videos (two lanes; default n // 4): 16- and 20-frame videos with caps of 4 .. 16 frames
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', d['config']['pass_ms'], d['frames_check']['ok'])"; }
for steps in 20 16; do
for b in 0 -3 -4 -6 -8 -10 -16 0; do
  echo "== --steps $steps KBE_DELIVERY_BATCH=$b"; KBE_DELIVERY_BATCH=$b timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | line
done
done
