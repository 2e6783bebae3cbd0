#!/bin/bash
# GPU box (round 4): the default bench line (and --steps 20) for prebuilt libraries (_variants/NAME.so ...), alternating: delivered, left in HBM, the crop kernel alone
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM, roofline %.4f, crop alone %.2f us, fill %.2f us, ok %s' % (d['value'], d['device_only']['value'], r['frac'], r['kernel_us']['crop_resize'], r['kernel_us']['fill'], d['frames_check']['ok']))"; }
for rep in 1 2; do
for lib in "$@"; do
  cp $R/_variants/$lib.so $R/ken-burns-effect_amd/csrc/libkbe_hip.so
  for args in "" "--steps 20 --warmup 5"; do
    echo "$lib [$args]: $(timeout 600 python bench.py --no-cpu-baseline $args 2>/dev/null | val)"
  done
done
done
