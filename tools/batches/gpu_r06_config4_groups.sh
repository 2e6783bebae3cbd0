#!/bin/bash
# GPU box, round 6: configs[4] (2048^2, 16.8 M points) by frames per scatter launch -- the rule "two per launch for dense clouds delivered to the
# host" dates from the blit hand-off (round 4: 446 / 422 / 395 us per delivered frame with 8 / 4 / 2); with the SDMA engine no copy kernel sits next to the launches
#   gpurun --timeout 1200 -- 'bash tools/batches/gpu_r06_config4_groups.sh r06b'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r06b}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for g in 2 4 8 3 6; do
  KBE_FILL_GROUP=$g timeout 900 python $R/bench.py --no-cpu-baseline --size 2048 --upsample 2 --steps 64 --warmup 8 2>> $OUT/bench.err | tail -1 > $OUT/bench_config4_group$g.json
done
for f in $OUT/bench_config4_group*.json; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    r = d.get('roofline') or {}
    print('delivered %.0f  in HBM %.0f  | roofline: %s frames per launch, %.1f us per frame, frac %.4f' % (d['value'], d['device_only']['value'], r.get('frames_per_launch'), r.get('us_per_frame', 0), r.get('frac', 0)))
except Exception as e:
    print('unreadable', e)
PY
done
tail -3 $OUT/bench.err
