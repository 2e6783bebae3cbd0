#!/bin/bash
# GPU box: tools/ahead_time.py for the shipped library and for variant builds (quoted sets of extra hipcc flags).
#   gpurun -- 'bash tools/gpu_ahead.sh "-DKBE_AHEAD_AT=1"'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
echo "== shipped"
timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -v Warning | tail -14
i=0
for flags in "$@"; do
  so=/tmp/libkbe_var_$i.so
  make -s -B -C $R/ken-burns-effect_amd/csrc EXTRA="$flags" OUT=$so || exit 1
  echo "== variant: $flags"
  KBE_LIB_PATH=$so timeout 600 python $R/tools/ahead_time.py 2>&1 | grep -v Warning | tail -14
  i=$((i+1))
done
