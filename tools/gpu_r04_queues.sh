#!/bin/bash
# GPU box (round 4): the dolly video (and, for comparison, the KBE video and the dense cloud) by hardware queues and lanes.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('delivered %.0f (lanes %s), left in HBM %.0f frames/s (lanes %s); frames ok %s' % (d['value'], d['config']['lanes'], d['device_only']['value'], d['config']['device_only_lanes'], d['frames_check']['ok']))"; }
for env in "" "GPU_MAX_HW_QUEUES=8 KBE_LANES=8" "GPU_MAX_HW_QUEUES=8 KBE_LANES=6" "GPU_MAX_HW_QUEUES=8 KBE_LANES=4" "GPU_MAX_HW_QUEUES=8 KBE_LANES=8 KBE_HOST_LANES=8" "GPU_MAX_HW_QUEUES=8 KBE_LANES=8 KBE_HOST_LANES=6"; do
  echo "== dolly [$env]: $(env $env timeout 600 python bench.py --dolly --steps 256 --warmup 64 --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/queues.txt
done
for env in "" "GPU_MAX_HW_QUEUES=8" "GPU_MAX_HW_QUEUES=8 KBE_LANES=8"; do
  echo "== kbe [$env]: $(env $env timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | line)" | tee -a $O/queues.txt
done
