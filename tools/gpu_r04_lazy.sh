#!/bin/bash
# GPU box (round 4): variants of how the tile launch gets its records' colours (_variants/NAME.so built here beforehand with
# tools/build_variants.sh): the scatter alone for each name given, then -- with $INSTALL in the library's place -- the parity
# suites, the bench line and the dense cloud's line.    INSTALL=m1f1 bash tools/gpu_r04_lazy.sh tree m1f1 m1f0
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in "$@" "$@"; do
  echo "== $v" | tee -a $O/lazy.txt
  KBE_LIB_PATH=$R/_variants/$v.so REPS=40 timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -E "^ ?(8|12) frame\(s\) per launch|max \|diff\| [2-9]" | tee -a $O/lazy.txt
done
cd $R
dense() { timeout 600 python bench.py --size 2048 --upsample 2 --steps 256 --warmup 64 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('dense: delivered %.0f, left in HBM %.0f frames/s; scatter %.2f us per frame (frac %.3f), one frame per launch %.2f; frames ok %s' % (d['value'], d['device_only']['value'], r['us_per_frame'], r['frac'], r['one_frame_per_launch']['us_per_frame'], d['frames_check']['ok']))"; }
echo "== dense, tree" | tee -a $O/lazy.txt; dense | tee -a $O/lazy.txt
if [ -n "$INSTALL" ]; then
  cp $R/_variants/$INSTALL.so $R/ken-burns-effect_amd/csrc/libkbe_hip.so
  echo "== installed: $INSTALL" | tee -a $O/lazy.txt
  timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_at_size.py tests/test_hip_reference.py -x -q -m gpu 2>&1 | tail -5 | tee -a $O/lazy.txt
  timeout 600 python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-1300 | tee -a $O/lazy.txt
  dense | tee -a $O/lazy.txt
fi
