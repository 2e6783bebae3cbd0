#!/bin/bash
# GPU box, round 6: the scatter's waves at different issue priorities by phase (s_setprio; variant libraries by tools/build_variants.sh)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for v in default $1 default; do
  envs="KBE_LIB_PATH=$R/_variants/$v.so"; [ $v = default ] && envs="KBE_NONE=1"
  echo "== $v"
  env $envs IDENTICAL=12 PATHS=75,1024 LAUNCH_FRAMES=12 REPS=60 SKIP_CHECK=1 timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -E "frame\(s\)|consecutive|rror"
  env $envs timeout 600 python $R/bench.py --no-cpu-baseline --device-only --steps 256 --warmup 64 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("frames/s left in HBM", round(d["value"]))'
done
