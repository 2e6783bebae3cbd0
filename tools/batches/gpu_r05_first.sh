#!/bin/bash
# GPU box (round 5, first call): the SDMA probe, the scatter launch on consecutive cameras (timing + tile counters), the shapes of a
# rank's share, and a bench line with the new `roofline.cameras`
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_first
mkdir -p $O
cd $R
echo "== sdma probe"; timeout 120 $R/_variants/sdma_probe > $O/sdma_probe.txt 2>&1; tail -40 $O/sdma_probe.txt
echo "== ahead_time (shipped library)"; REPS=40 timeout 900 python tools/ahead_time.py > $O/ahead_time.txt 2>&1; grep -E "per launch|consecutive|max \|diff\| [2-9]" $O/ahead_time.txt
echo "== ahead_time (tile counters)"; KBE_LIB_PATH=$R/_variants/stats.so IDENTICAL=0 REPS=24 timeout 900 python tools/ahead_time.py > $O/ahead_stats.txt 2>&1; grep -E "consecutive" $O/ahead_stats.txt
echo "== shard shapes"; timeout 900 python tools/shard_shapes.py > $O/shard_shapes.txt 2>&1; grep -E "video" $O/shard_shapes.txt
echo "== bench --steps 20"; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; tail -c 600 $O/bench_steps20.err
python3 - <<PY
import json
d=json.loads(open('$O/bench_steps20.json').read().strip().split('\n')[-1])
r=d['roofline']
print('value', d['value'], 'device_only', d['device_only']['value'], 'frac', r['frac'], r['cameras'])
print('by_path', {k:(v['us_per_frame'], round(v['frac'],3)) for k,v in r['by_path'].items()}, 'identical', r['identical_cameras'])
PY
