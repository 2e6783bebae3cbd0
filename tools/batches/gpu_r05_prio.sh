#!/bin/bash
# GPU box (round 5): the lanes beside the caller's stream as LOW-priority HIP streams (the default since this measurement) against torch's own streams (KBE_LANE_PRIORITY=0)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', d['config']['pass_ms'], round(d['device_only']['value'],1), 'in HBM', d['frames_check']['ok'])"; }
for steps in 20 16 75 1024; do
for pr in 0 1 0 1 0 1; do
  echo "== --steps $steps KBE_LANE_PRIORITY=$pr"; KBE_LANE_PRIORITY=$pr timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | line
done
done
timeout 900 python -m pytest tests -m gpu -x -q -k "video or sdma or handoff or deliver or shard or rank or back_to_back" 2>&1 | tail -2
