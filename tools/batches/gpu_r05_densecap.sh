#!/bin/bash
# GPU box (round 5): configs[4] (16.8 M points on 2048^2) with more records per tile in LDS and fewer workgroups per CU: 736 x 5 (the build) / 1020 x 4 / 1472 x 3
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['value'],1), 'delivered', round(d['device_only']['value'],1), 'left in HBM; the dense launch', r.get('us_per_frame'), 'us per frame ->', round(r.get('frac'),4), d['frames_check']['ok'])"; }
for v in ${VARIANTS:-base cap1020 cap1472 base cap1020 cap1472}; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so timeout 600 python bench.py --no-cpu-baseline --size 2048 --upsample 2 --steps 64 --warmup 8 2>/dev/null | line
done
