#!/bin/bash
# GPU box (round 4, first batch): the new division's self-test and the fused-route tests on the tree's library; the scatter alone
# on a stream for the variant libraries of tools/build_variants.sh; the co-issue pairs of tools/valu_rate.hip; per-kernel
# durations of configs[4] (2048^2 from 16.8 M points) and of the dolly video on one lane, both scatter routes.
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R && timeout 600 python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "division or pipelined or fused or ahead" 2>&1 | tail -3 | tee $O/a_tests.txt
cd /tmp && export TMPDIR=/tmp
bash $R/tools/gpu_r04_variants.sh "$@" 2>&1 | grep -v "amdgpu.ids\|MIOpen" | tee $O/a_variants.txt
hipcc --offload-arch=gfx950 -O2 $R/tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate from=64 2 4 8 > $O/a_valu_pairs.txt 2>&1
tail -4 $O/a_valu_pairs.txt
for env in "SIZE=2048 UPSAMPLE=2 CLOUD=raw" "DOLLY=1"; do
  for fused in 0 1; do
    tag=$(echo "$env fused$fused" | tr ' =' '__')
    rm -rf /tmp/kt
    env $env KBE_FUSED=$fused KBE_LANES=1 FRAMES=32 REPS=2 timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt -o t --output-format csv -- python $R/tools/throughput.py > /tmp/kt.log 2>&1
    tail -1 /tmp/kt.log
    cp /tmp/kt/t_kernel_stats.csv $O/a_kstats_$tag.csv 2>/dev/null
    head -8 $O/a_kstats_$tag.csv | cut -c1-160
  done
done
