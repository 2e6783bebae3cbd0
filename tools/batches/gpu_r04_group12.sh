#!/bin/bash
# GPU box (round 4): delivered videos with 8 (default) against 12 frames per launch of the fused scatter, at the bench's three lengths
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.0f delivered (%.3f ms per pass, lanes %s), %.0f left in HBM, roofline %.3f at %d frames per launch, ok %s' % (d['value'], d['config']['pass_ms']['median'], d['config']['lanes'], d['device_only']['value'], d['roofline']['frac'], d['roofline']['frames_per_launch'], d['frames_check']['ok']))"; }
for rep in 1 2; do
for g in "" 12; do
  for args in "" "--steps 20 --warmup 20" "--steps 75 --warmup 20"; do
    echo "group ${g:-default} [$args]: $(env ${g:+KBE_FILL_GROUP=$g} timeout 600 python bench.py --no-cpu-baseline $args 2>/dev/null | val)"
  done
done
done
