#!/bin/bash
# GPU box (round 4): MIOpen find modes under torch.backends.cudnn.benchmark on a box with no find-db: first call and steady state
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for net in plain partial; do
  for combo in "MIOPEN_FIND_MODE=FAST BENCHMARK=1 CL=0" "MIOPEN_FIND_MODE=FAST BENCHMARK=1 CL=0" "MIOPEN_FIND_MODE=FAST BENCHMARK=0 CL=0" "MIOPEN_FIND_MODE=DYNAMIC_HYBRID BENCHMARK=1 CL=0" "MIOPEN_FIND_MODE=FAST BENCHMARK=1 CL=0" "MIOPEN_FIND_MODE=FAST BENCHMARK=0 CL=0"; do
    echo "== $net $combo: $(env $combo NET=$net timeout 600 python $R/tools/cnn_time.py 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tr '\n' ' ')"
  done
done 2>&1 | tee $O/nets2_cnn_time.txt
