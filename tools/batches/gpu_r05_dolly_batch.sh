#!/bin/bash
# GPU box (round 5): where the rendering binds (all lanes deliver), transfer groups of ONE scatter launch's frames against groups of up to 32
# (KBE_DELIVERY_BATCH=-32: round 5's earlier shape) -- dolly, a raw cloud, 2048^2 raw, configs[4]; and the default workload, which keeps two lanes
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', round(d['device_only']['value'],1), 'left in HBM', d['frames_check']['ok'], 'lanes', d['config']['lanes'])"; }
for args in "--dolly --steps 256 --warmup 32" "--cloud raw --steps 256 --warmup 32" "--size 2048 --cloud raw --steps 128 --warmup 16" "--size 2048 --upsample 2 --steps 64 --warmup 8" "--steps 256 --warmup 32" "--steps 75 --warmup 8"; do
  for b in -32 0 -32 0; do
    echo "== $args KBE_DELIVERY_BATCH=$b"; KBE_DELIVERY_BATCH=$b timeout 300 python bench.py --no-cpu-baseline $args 2>/dev/null | line
  done
done
