#!/bin/bash
# GPU box (round 5): the dolly fill's cooperative creeping rays (KBE_FILL_COOP_LANES variants), shard shapes incl. dealt runs (SDMA hand-off), GPU suite
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_fourth
mkdir -p $O
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM (%.1f us per frame), ok %s; roofline %s: %.1f us per frame -> %.4f, worst of eight cameras %s' % (d['value'] or -1, d['device_only']['value'], d['device_only']['ms_per_step']*1e3, d['frames_check']['ok'], r['kernel'][:40], r['us_per_frame'], r['frac'], r.get('us_worst_of_eight_cameras')))"; }
for v in coop0 coop4 coop2 coop8 coop0 coop4; do
  echo "$v [--dolly]: $(KBE_LIB_PATH=$R/_variants/$v.so timeout 900 python bench.py --no-cpu-baseline --dolly --steps 256 --warmup 32 2>/dev/null | tee $O/bench_dolly_$v.json | val)"
done
echo "== shard shapes (sdma)"; KBE_HANDOFF=sdma timeout 900 python tools/shard_shapes.py 2>&1 | tee $O/shard_shapes_sdma.txt | grep -E "video"
echo "== gpu tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
