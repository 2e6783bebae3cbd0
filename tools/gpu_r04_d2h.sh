#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in "X=1" "GPU_BLIT_ENGINE_TYPE=1" "GPU_BLIT_ENGINE_TYPE=2" "GPU_BLIT_ENGINE_TYPE=3" "GPU_FORCE_BLIT_COPY_SIZE=0" "HSA_ENABLE_SDMA=1 GPU_FORCE_BLIT_COPY_SIZE=0" "DEBUG_CLR_LIMIT_BLIT_WG=4" "DEBUG_CLR_LIMIT_BLIT_WG=64" "GPU_PINNED_MIN_XFER_SIZE=0"; do
  rm -rf /tmp/dp
  env $e timeout 120 rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/dp -o t --output-format csv -- python $R/tools/d2h_env_probe.py 2>/dev/null | grep "copy alone"
  echo "   [$e] copyBuffer kernels: $(grep -c copyBuffer /tmp/dp/t_kernel_trace.csv 2>/dev/null), memory-copy trace rows: $(wc -l < /tmp/dp/t_memory_copy_trace.csv 2>/dev/null) $(cut -d, -f3-5 /tmp/dp/t_memory_copy_trace.csv 2>/dev/null | sort | uniq -c | head -4 | tr '\n' ';')"
done 2>&1 | tee $O/l_d2h_env.txt
