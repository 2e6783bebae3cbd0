#!/bin/bash
# GPU box (round 4): the Inpaint networks at 1024^2 by memory format and MIOpen mode (tools/cnn_time.py), then bench.py --pipeline
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for net in plain partial; do
  for combo in "MIOPEN_FIND_MODE=FAST BENCHMARK=0 CL=0" "MIOPEN_FIND_MODE=FAST BENCHMARK=0 CL=1" "BENCHMARK=1 CL=0" "BENCHMARK=1 CL=1" "BENCHMARK=1 CL=1" "MIOPEN_FIND_MODE=NORMAL BENCHMARK=1 CL=0"; do
    echo "== $net $combo: $(env $combo NET=$net timeout 600 python $R/tools/cnn_time.py 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tr '\n' ' ')"
  done
done 2>&1 | tee $O/nets_cnn_time.txt
cd $R
MIOPEN_FIND_MODE=FAST timeout 600 python bench.py --pipeline --steps 10 --warmup 2 2>/dev/null | tail -1 > $O/nets_pipeline_default.json
timeout 900 python bench.py --pipeline --steps 10 --warmup 2 --miopen-find 2>/dev/null | tail -1 > $O/nets_pipeline_find.json
python - <<P
import json
for f in ('default', 'find'):
    d = json.loads(open('$O/nets_pipeline_%s.json' % f).read())
    print(f, d['ms_per_step'], d['stages_ms'], d['partial_inpaint_1024'])
P
ls ~/.config/miopen 2>/dev/null | head; du -sh ~/.config/miopen ~/.cache/miopen 2>/dev/null
