#!/bin/bash
# GPU box (round 5): the dolly zoom on the fused route now that its lists hold 2048 sub-blocks (round 3: 147 us per frame against the bucket route's 97)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM (%.1f us per frame), route %s lanes %s, ok %s; roofline %s %.1f us per frame -> %.4f' % (d['value'] or -1, d['device_only']['value'], d['device_only']['ms_per_step']*1e3, d['config']['scatter_route'], d['config']['lanes'], d['frames_check']['ok'], r['kernel'][:24], r['us_per_frame'], r['frac']))"; }
for e in "KBE_FUSED=auto" "KBE_FUSED=1" "KBE_FUSED=1 KBE_FILL_GROUP=4" "KBE_FUSED=1 KBE_FILL_GROUP=8" "KBE_FUSED=1 KBE_FILL_GROUP=12" "KBE_FUSED=auto" "KBE_FUSED=1"; do
  echo "$e [--dolly]: $(env $e timeout 900 python bench.py --no-cpu-baseline --dolly --steps 256 --warmup 32 2>/dev/null | val)"
done
for e in "KBE_FUSED=auto" "KBE_FUSED=1"; do echo "$e [--dolly --size 512]: $(env $e timeout 900 python bench.py --no-cpu-baseline --dolly --size 512 --steps 256 --warmup 32 2>/dev/null | val)"; done
