#!/bin/bash
# GPU box (round 4): the colours fetched for the records only (KBE_LAZY_COLOURS=1, _variants/lazy.so built here beforehand)
# against the tree: the scatter alone, then -- with the variant in the library's place -- the parity suites and the bench lines.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in tree lazy tree lazy; do
  echo "== $v"
  KBE_LIB_PATH=$R/_variants/$v.so REPS=40 timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -E "frame\(s\) per launch|max \|diff\| [2-9]" | tee -a $O/lazy.txt
done
cd $R
echo "== dense, tree"; timeout 600 python bench.py --size 2048 --upsample 2 --steps 256 --warmup 64 --no-cpu-baseline 2>&1 | tail -1 | tee -a $O/lazy.txt
cp $R/_variants/lazy.so $R/ken-burns-effect_amd/csrc/libkbe_hip.so
cd $R
timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_at_size.py tests/test_hip_reference.py -x -q -m gpu 2>&1 | tail -5 | tee -a $O/lazy.txt
timeout 600 python bench.py 2>&1 | tail -1 | tee -a $O/lazy.txt
echo "== dense, lazy"; timeout 600 python bench.py --size 2048 --upsample 2 --steps 256 --warmup 64 --no-cpu-baseline 2>&1 | tail -1 | tee -a $O/lazy.txt
