#!/bin/bash
# GPU box (round 4): bench lines of the other configurations for two prebuilt libraries (_variants/$1.so, _variants/$2.so), alternating
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM, roofline %.4f (%.2f us per frame, %s per launch), ok %s' % (d['value'], d['device_only']['value'], r['frac'], r['us_per_frame'], r.get('frames_per_launch'), d['frames_check']['ok']))"; }
for rep in 1 2; do
for lib in "$@"; do
  cp $R/_variants/$lib.so $R/ken-burns-effect_amd/csrc/libkbe_hip.so
  for args in "--size 2048 --upsample 2 --steps 64 --warmup 8" "--dolly --steps 256 --warmup 32" ${EXTRA_ARGS:+"$EXTRA_ARGS"}; do
    echo "$lib [$args]: $(timeout 600 python bench.py --no-cpu-baseline $args 2>/dev/null | val)"
  done
done
done
