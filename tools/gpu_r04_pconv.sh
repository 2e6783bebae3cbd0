#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R && MIOPEN_FIND_MODE=FAST timeout 900 python -m pytest tests/test_hip_parity.py tests/test_hip_reference.py -x -q -m gpu -k "pconv or partial" 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
for combo in "MIOPEN_FIND_MODE=FAST BENCHMARK=0" "BENCHMARK=1" "MIOPEN_FIND_MODE=FAST BENCHMARK=0"; do
  echo "== partial $combo: $(env $combo NET=partial CL=0 timeout 600 python $R/tools/cnn_time.py 2>&1 | grep -v "MIOpen\|amdgpu.ids" | tr '\n' ' ')"
done 2>&1 | tee $O/n_pconv_fused.txt
