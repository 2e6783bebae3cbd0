#!/bin/bash
# GPU box (round 5): cooperative rays in the table fill, relaxed (any walking ray when few lanes walk): variants 0 / 4 / 8 / 16 lanes
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM (%.1f us per frame), ok %s; fill %d per launch: %.1f us per frame -> %.4f (worst %s); one per launch %s' % (d['value'] or -1, d['device_only']['value'], d['device_only']['ms_per_step']*1e3, d['frames_check']['ok'], r.get('frames_per_launch',1), r['us_per_frame'], r['frac'], r.get('us_worst_of_eight_cameras'), r.get('one_frame_per_launch',{}).get('us_per_frame')))"; }
for rep in 1 2; do for v in coop0 coop4 coop8 coop16; do
  echo "$v [--dolly]: $(KBE_LIB_PATH=$R/_variants/$v.so timeout 900 python bench.py --no-cpu-baseline --dolly --steps 256 --warmup 32 2>/dev/null | val)"
done; done
for e in "" "-DKBE_FILL_COOP_LANES=0"; do echo "== fill stats [$e]"; EXTRA="$e" timeout 900 python tools/fill_stats.py 2>&1 | grep -E "ray ends|^holes" | tail -2 | cut -c1-330; done
timeout 2400 python -m pytest tests -x -q -m gpu -k "fill or hole or dolly or schedule" 2>&1 | tail -3
