#!/bin/bash
# GPU box: fused vs bucket path: kernel medians (1 lane), throughput (device-only / host), default bench (dev aid, round 2)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/${OUT:-r02j}
mkdir -p $O
cd $R
for fused in 1 0; do
  echo "==== KBE_FUSED=$fused" >> $O/log.txt
  KBE_FUSED=$fused LANES=1 FRAMES=65 OUT=${OUT:-r02j}/f$fused bash tools/gpu_timeline.sh >> $O/log.txt 2>&1
  KBE_FUSED=$fused FRAMES=512 REPS=5 python tools/throughput.py 2>/dev/null | tail -1 >> $O/log.txt
  KBE_FUSED=$fused KBE_LANES=1 FRAMES=512 REPS=5 python tools/throughput.py 2>/dev/null | tail -1 >> $O/log.txt
  KBE_FUSED=$fused HOST=1 FRAMES=512 REPS=5 python tools/throughput.py 2>/dev/null | tail -1 >> $O/log.txt
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
cat $O/log.txt; head -c 3000 $O/bench.json
