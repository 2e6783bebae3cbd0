#!/bin/bash
# GPU box (round 5): dilations of the fill's distance table (KBE_DIST_CAP 15 / 11 / 8) on the dolly bench
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM (%.1f us per frame), ok %s; fill %d per launch: %.1f us per frame -> %.4f' % (d['value'] or -1, d['device_only']['value'], d['device_only']['ms_per_step']*1e3, d['frames_check']['ok'], r.get('frames_per_launch',1), r['us_per_frame'], r['frac']))"; }
for rep in 1 2; do for v in "" dist11 dist8; do
  echo "${v:-shipped (15)} [--dolly]: $(KBE_LIB_PATH=${v:+$R/_variants/$v.so} timeout 900 python bench.py --no-cpu-baseline --dolly --steps 256 --warmup 32 2>/dev/null | val)"
done; done
