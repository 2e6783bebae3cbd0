#!/bin/bash
# GPU box: throughput (default lanes and 1 lane) for the three workloads, per variant build (dev aid)
#   gpurun -- 'bash tools/gpu_cases.sh "" "-DKBE_PROBE_FILL_GLOBAL"'
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  so=/tmp/libkbe_case_$i.so
  make -s -B -C $R/ken-burns-effect_amd/csrc EXTRA="$flags" OUT=$so || exit 1
  echo "==== variant: ${flags:-(default)}"
  for c in ${CASES:-"CLOUD=inpaint,DOLLY=0" "CLOUD=raw,DOLLY=0" "CLOUD=raw,DOLLY=1"}; do
    echo "== $c"
    env ${c//,/ } KBE_LIB_PATH=$so FRAMES=${FRAMES:-64} REPS=3 python $R/tools/throughput.py 2>/dev/null | tail -1
    env ${c//,/ } KBE_LIB_PATH=$so KBE_LANES=1 FRAMES=${FRAMES:-64} REPS=3 python $R/tools/throughput.py 2>/dev/null | tail -1
  done
  i=$((i+1))
done
