#!/bin/bash
# GPU box (round 4, second batch): the whole GPU suite; variant libraries alone on a stream; the wave timeline of the tree's
# kernel; the headline bench line (short).
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee $O/b_tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/gpu_r04_variants.sh "$@" 2>&1 | grep -v "amdgpu.ids\|MIOpen" | grep "variant\|^ 8 \|^12 \|^ 4 " | tee $O/b_variants.txt
FRAMES=8 timeout 600 python $R/tools/frame_probe.py 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tail -22 | tee $O/b_frame_probe_8.txt
cd $R && timeout 900 python bench.py --steps 256 --warmup 64 > $O/b_bench.json 2> $O/b_bench.err; tail -3 $O/b_bench.err; python - <<P
import json
d = json.loads(open('$O/b_bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['device_only']['value'], {k: d['roofline'][k] for k in ('frac', 'us_per_frame', 'frames_per_launch')}, d.get('frames_check'))
P
