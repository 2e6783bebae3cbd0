#!/bin/bash
# GPU box (round 5): the dolly bench and the fill's GPU tests on the current tree
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_dolly
mkdir -p $O
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM (%.1f us per frame), ok %s; roofline %s: %.1f us per frame -> %.4f, worst of eight cameras %s' % (d['value'] or -1, d['device_only']['value'], d['device_only']['ms_per_step']*1e3, d['frames_check']['ok'], r['kernel'][:40], r['us_per_frame'], r['frac'], r.get('us_worst_of_eight_cameras')))"; }
for rep in 1 2; do echo "--dolly: $(timeout 900 python bench.py --no-cpu-baseline --dolly --steps 256 --warmup 32 2>/dev/null | tee $O/bench_dolly_$rep.json | val)"; done
echo "--dolly --size 512: $(timeout 900 python bench.py --no-cpu-baseline --dolly --size 512 --steps 256 --warmup 32 2>/dev/null | val)"
timeout 900 python tools/fill_stats.py 2>&1 | tee $O/fill_stats.txt | grep "^holes" | tail -3
timeout 2400 python -m pytest tests -x -q -m gpu -k "fill or hole or dolly or schedule" 2>&1 | tail -3
