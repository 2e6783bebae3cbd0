#!/bin/bash
# GPU box (round 5): the dolly zoom by route on the final tree (the fused route is the default for zoom-outs now), and the GPU suite
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_dolly5
mkdir -p $O
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM (%.1f us per frame), route %s lanes %s, ok %s; roofline %s, %s per launch: %.1f us per frame -> %.4f; one per launch %s' % (d['value'] or -1, d['device_only']['value'], d['device_only']['ms_per_step']*1e3, d['config']['scatter_route'], d['config']['lanes'], d['frames_check']['ok'], r['kernel'][:24], r.get('frames_per_launch',1), r['us_per_frame'], r['frac'], r.get('one_frame_per_launch',{}).get('us_per_frame')))"; }
for size in 1024 512 2048; do for e in "KBE_FUSED=auto" "KBE_FUSED=0"; do
  echo "$e [--dolly --size $size]: $(env $e timeout 900 python bench.py --no-cpu-baseline --dolly --size $size --steps 256 --warmup 32 2>/dev/null | tee $O/bench_dolly_${size}_${e#KBE_FUSED=}.json | val)"
done; done
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
