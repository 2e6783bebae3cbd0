#!/bin/bash
# GPU box (round 5): the SDMA hand-off on one engine against two engines taking the groups in turn
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.0f delivered (%.3f ms per pass, pcie %.1f GB/s), ok %s' % (d['value'] or -1, d['config']['pass_ms']['median'], d['pcie']['achieved'], d['frames_check']['ok']))"; }
for rep in 1 2; do for lib in "" $R/_variants/one_engine.so; do for args in "--steps 20 --warmup 5" "--steps 75 --warmup 20" "--steps 16 --warmup 5"; do
  echo "${lib:+one engine}${lib:-two engines} [$args]: $(KBE_LIB_PATH=$lib timeout 600 python bench.py --no-cpu-baseline $args 2>/dev/null | val)"
done; done; done
echo "two engines [--steps 1024]: $(timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | val)"
bash tools/gpu_r05_timeline.sh 2>&1 | grep -E "COPY|first copy"
