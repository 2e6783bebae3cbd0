#!/bin/bash
# GPU box (round 5): the frame loop of BASELINE configs[1] -- a 64-frame 512^2 delivered video -- by lanes and transfer-group cap
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', d['config']['pass_ms'], 'lanes', d['config']['lanes'], round(d['device_only']['value'],1), 'in HBM', d['frames_check']['ok'])"; }
for spec in "" "KBE_HOST_LANES=2" "KBE_HOST_LANES=2 KBE_DELIVERY_BATCH=-32" "KBE_HOST_LANES=3" "KBE_HOST_LANES=4 KBE_DELIVERY_BATCH=-4" "KBE_HOST_LANES=4 KBE_DELIVERY_BATCH=-8" "KBE_HOST_LANES=4 KBE_DELIVERY_BATCH=-16" "KBE_HOST_LANES=4 KBE_DELIVERY_BATCH=-32" "KBE_HOST_LANES=2 KBE_DELIVERY_BATCH=-8" ""; do
  echo "== --size 512 --steps 64 $spec"; env $spec timeout 300 python bench.py --no-cpu-baseline --size 512 --steps 64 --warmup 8 2>/dev/null | line
done
