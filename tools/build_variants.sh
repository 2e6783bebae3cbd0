#!/bin/bash
# Here (no GPU): variant libraries for tools/batches/gpu_r04_variants.sh, one per "name=flags" argument, into _variants/ (git-ignored; they
# travel to the GPU box with the snapshot).   bash tools/build_variants.sh "head=-DKBE_XCD_ROT=0 -DKBE_LATE_ARGS=0" "tree="
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/_variants
for spec in "$@"; do
  name=${spec%%=*}; flags=${spec#*=}
  ( make -s -B -C $R/ken-burns-effect_amd/csrc EXTRA="$flags -Rpass-analysis=kernel-resource-usage" OUT=$R/_variants/$name.so 2>&1 \
      | grep -A10 "19k_frame_group_aheadE" | grep -E "VGPRs:|Spill|LDS Size|Occupancy" | sed 's/.*remark: *//; s/ \[-Rpass.*//' | tr '\n' ';'; echo " <- $name ($flags)" ) &
  while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
done
wait
