#!/bin/bash
# GPU box (round 5): the SDMA hand-off's signals as GPU-only signals (no host interrupt behind them; value address from hsa_amd_signal_value_pointer)
# against default signals whose value is found through amd_signal_t (base)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', round(d['config']['pass_ms']['median'],4), 'ms per pass', d['frames_check']['ok'])"; }
for v in base gpusig base gpusig base gpusig; do
  for steps in 20 75; do
    echo "== $v --steps $steps"; KBE_LIB_PATH=$R/_variants/$v.so timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | line
  done
done
timeout 900 python -m pytest tests -m gpu -x -q -k "video or sdma or handoff or deliver" 2>&1 | tail -3
