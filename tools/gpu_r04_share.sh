#!/bin/bash
# GPU box (round 4): the shared candidate lists (_variants/share.so in the library's place) on delivered videos of several shapes
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.0f delivered (%.3f ms per pass, lanes %s), %.0f left in HBM, ok %s' % (d['value'], d['config']['pass_ms']['median'], d['config']['lanes'], d['device_only']['value'], d['frames_check']['ok']))"; }
for lib in noshare share; do
  cp $R/_variants/$lib.so $R/ken-burns-effect_amd/csrc/libkbe_hip.so
  for args in "--steps 20 --warmup 20" "--steps 75 --warmup 20" "--size 2048 --upsample 2 --steps 64 --warmup 8" "--size 512 --steps 256 --warmup 64" "--cloud raw --steps 256 --warmup 64"; do
    echo "$lib [$args]: $(timeout 600 python bench.py --no-cpu-baseline $args 2>/dev/null | val)"
  done
  for e in "KBE_HOST_LANES=1" "KBE_HOST_LANES=2" "KBE_FILL_GROUP=1" "KBE_FILL_GROUP=4"; do
    echo "$lib dense [$e]: $(env $e timeout 600 python bench.py --no-cpu-baseline --size 2048 --upsample 2 --steps 64 --warmup 8 2>/dev/null | val)"
  done
done
cp $R/_variants/share.so $R/ken-burns-effect_amd/csrc/libkbe_hip.so
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -2
cd /tmp
for v in noshare share noshare share; do echo "== $v"; KBE_LIB_PATH=$R/_variants/$v.so REPS=40 timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -E "^ ?(4|8|12) frame\(s\) per launch|max \|diff\| [2-9]"; done
