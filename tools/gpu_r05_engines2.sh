#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.0f delivered (%.3f ms per pass, pcie %.1f GB/s), ok %s' % (d['value'] or -1, d['config']['pass_ms']['median'], d['pcie']['achieved'], d['frames_check']['ok']))"; }
for rep in 1 2 3; do for args in "--steps 20 --warmup 5" "--steps 75 --warmup 20"; do
  echo "[$args]: $(timeout 600 python bench.py --no-cpu-baseline $args 2>/dev/null | val)"
done; done
bash tools/gpu_r05_timeline.sh 2>&1 | tail -62
