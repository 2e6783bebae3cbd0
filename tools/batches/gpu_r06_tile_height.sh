#!/bin/bash
# GPU box, round 6: the fused scatter on 32 x 8 tiles (KBE_TILE_H=8: one pixel per thread, ~300 records per tile -- no second rounds at one
# point per pixel, more halo per pixel) against 32 x 16; variant libraries built by tools/build_variants.sh
#   gpurun --timeout 2400 -- 'bash tools/batches/gpu_r06_tile_height.sh r06t "th8 th8c384 th8c384w7 th8c320w8 wide0"'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r06t}
mkdir -p $OUT
for v in $2; do
  cd $R
  echo "== $v: parity: $(KBE_LIB_PATH=$R/_variants/$v.so timeout 900 python -m pytest tests/test_hip_parity.py -q -x -k 'frames_match_oracle or every_instantiation or full_size_frame or pile_up or degenerate or random_small or fused_scatter_equals or cropped_frames or hole_fill_schedules' 2>&1 | tail -1)"
done
cd /tmp && export TMPDIR=/tmp
for v in default $2 default; do
  envs="KBE_LIB_PATH=$R/_variants/$v.so"; [ $v = default ] && envs="KBE_NONE=1"
  echo "== $v"
  env $envs IDENTICAL=12 PATHS=75,1024 LAUNCH_FRAMES=12 REPS=60 SKIP_CHECK=1 timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -E "frame\(s\)|consecutive|rror"
  env $envs timeout 600 python $R/bench.py --no-cpu-baseline --device-only --steps 256 --warmup 64 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("frames/s left in HBM", round(d["value"]))'
done 2>&1 | tee $OUT/tile_height.txt
