#!/bin/bash
# GPU box: for each variant (a quoted set of extra hipcc flags) build the library into /tmp, trace 33 frames
# and print the per-kernel median durations (dev aid).   gpurun -- 'bash tools/gpu_variants.sh "" "-DX=2"'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  so=/tmp/libkbe_var_$i.so
  make -s -B -C $R/ken-burns-effect_amd/csrc EXTRA="$flags" OUT=$so || exit 1
  rm -rf /tmp/t$i
  KBE_LANES=1 KBE_LIB_PATH=$so FRAMES=${FRAMES:-33} timeout 600 rocprofv3 --kernel-trace -d /tmp/t$i -o t --output-format csv -- python $R/tools/frame_once.py > /tmp/t$i.log 2>&1 || tail -5 /tmp/t$i.log
  echo "== variant: ${flags:-(default)}"
  python $R/tools/kernel_times.py /tmp/t$i/t_kernel_trace.csv
  KBE_LIB_PATH=$so python $R/tools/throughput.py 2>/dev/null | tail -1
  i=$((i+1))
done
