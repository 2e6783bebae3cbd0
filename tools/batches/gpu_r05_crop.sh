#!/bin/bash
# GPU box (round 5): k_crop_resize_u8 with tiles of 64 x 16 / 32 / 64 walking down their rows against round 4's 64 x 8 (base)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_crop
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_hip_parity.py -q -x -k "crop" 2>&1 | tail -3
for v in ${VARIANTS:-base crop16 crop32 crop64 base crop16 crop32 crop64}; do
  echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so timeout 300 python tools/crop_time.py 2>&1 | tee -a $O/time_$v.txt | tail -2
done
for v in ${VARIANTS_BENCH:-base crop32 crop16 base crop32}; do
  echo "== bench $v"; KBE_LIB_PATH=$R/_variants/$v.so timeout 300 python bench.py --no-cpu-baseline --steps 256 --warmup 32 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', round(d['device_only']['value'],1), 'left in HBM', d['frames_check']['ok'])"
done
