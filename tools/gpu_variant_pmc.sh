#!/bin/bash
# GPU box: for each variant (a quoted set of extra hipcc flags) build the library into /tmp, collect one PMC pass
# ($PMC, default instruction counts) over 9 frames and print the per-kernel means (dev aid).
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
PMC=${PMC:-"SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS"}
cd /tmp && export TMPDIR=/tmp
i=0
for flags in "$@"; do
  so=/tmp/libkbe_var_$i.so
  make -s -B -C $R/ken-burns-effect_amd/csrc EXTRA="$flags" OUT=$so || exit 1
  rm -rf /tmp/v$i
  KBE_LIB_PATH=$so FRAMES=9 timeout 600 rocprofv3 --pmc $PMC -d /tmp/v$i -o c --output-format csv -- python $R/tools/frame_once.py > /tmp/v$i.log 2>&1 || tail -5 /tmp/v$i.log
  echo "== variant: ${flags:-(default)}"
  python $R/tools/pmc_kernels.py /tmp/v$i/c_counter_collection.csv | grep -E "${KERNELS:-k_frame}"
  i=$((i+1))
done
