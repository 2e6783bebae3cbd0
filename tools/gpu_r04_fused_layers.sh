#!/bin/bash
# GPU box (round 4): the plain networks with their element-wise passes fused (KBE_FUSED_LAYERS=1, the default) and as stock modules (0):
# the whole 512^2 video (bench.py --pipeline: the first process tunes MIOpen once), the 1024^2 Inpaint forward, the GPU time by kernel.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.2f ms per 512^2 video; stages %s; partial 1024^2 %s' % (d['ms_per_step'], json.dumps(d['config'].get('stages_ms', d.get('stages_ms'))), json.dumps({k: v for k, v in (d.get('partial_inpaint_1024') or {}).items() if k.endswith('_ms')})))"; }
python $R/bench.py --pipeline --steps 2 --warmup 1 > /dev/null 2>&1      # tunes once
for f in 1 0 1 0; do
  echo "KBE_FUSED_LAYERS=$f: $(KBE_FUSED_LAYERS=$f timeout 600 python $R/bench.py --pipeline --steps 10 --warmup 2 2>/dev/null | line)" | tee -a $O/fused_layers.txt
done
BENCHMARK=1 python $R/tools/cnn_time.py > /dev/null 2>&1                  # the find step once for the 1024^2 forward
for f in 1 0 1 0; do
  echo "KBE_FUSED_LAYERS=$f Inpaint 1024^2: $(KBE_FUSED_LAYERS=$f BENCHMARK=0 timeout 600 python $R/tools/cnn_time.py 2>/dev/null | head -1)" | tee -a $O/fused_layers.txt
done
rm -rf /tmp/pt; rocprofv3 --kernel-trace -d /tmp/pt -o t --output-format csv -- python $R/tools/pipeline_trace.py > /tmp/pt.log 2>&1
python $R/tools/pipeline_trace.py /tmp/pt/t_kernel_trace.csv | head -24 | tee -a $O/fused_layers.txt
