#!/bin/bash
# GPU box (round 5): round 4's compile-time knobs once more, now on real cameras (the lists are longer there): lazy colours, splat look-ahead, units up front
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for rep in 1 2 3; do for v in base lazy1 depth3 units2; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so IDENTICAL=0 SKIP_CHECK=1 PATHS=75,1024 LAUNCH_FRAMES=12 REPS=60 timeout 600 python tools/ahead_time.py 2>&1 | grep -E "consecutive"
done; done
( time timeout 600 python bench.py --steps 20 --warmup 5 > /tmp/b20.json 2>/dev/null ) 2>&1 | grep real; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
