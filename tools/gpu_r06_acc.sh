#!/bin/bash
# GPU box, round 6: the tile launch without records (KBE_FUSED_CAP=acc: frame_body_acc) against the lean / roomy / dense builds --
# the GPU suite under either, the scatter's launch alone on consecutive cameras, configs[4].
#   gpurun --timeout 1500 -- 'bash tools/gpu_r06_acc.sh r06a'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r06a}
mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -5 > $OUT/suite_default.txt
KBE_FUSED_CAP=acc timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -60 > $OUT/suite_acc.txt
cd /tmp && export TMPDIR=/tmp
for cap in default acc default acc; do
  envs="KBE_FUSED_CAP=$cap"; [ $cap = default ] && envs="KBE_NONE=1"
  env $envs IDENTICAL=12 PATHS=75,20 LAUNCH_FRAMES=12 REPS=60 timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -E "frame|consecutive|Error|error" >> $OUT/scatter_$cap.txt
done
for cap in default acc; do
  envs="KBE_FUSED_CAP=$cap"; [ $cap = default ] && envs="KBE_NONE=1"
  env $envs timeout 900 python $R/bench.py --no-cpu-baseline --size 2048 --upsample 2 --steps 64 --warmup 8 2>> $OUT/bench.err | tail -1 > $OUT/bench_config4_$cap.json
  env $envs timeout 900 python $R/bench.py --no-cpu-baseline --steps 75 --warmup 20 2>> $OUT/bench.err | tail -1 > $OUT/bench_steps75_$cap.json
done
tail -3 $OUT/suite_default.txt; tail -40 $OUT/suite_acc.txt; cat $OUT/scatter_default.txt; echo; cat $OUT/scatter_acc.txt; for f in $OUT/bench_*.json; do echo $f; python - "$f" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1]))
    print(d['value'], d.get('device_only'), json.dumps(d.get('roofline'))[:400])
except Exception as e:
    print('unreadable', e)
PY
done
tail -5 $OUT/bench.err
