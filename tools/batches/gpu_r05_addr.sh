#!/bin/bash
# GPU box (round 5): is the scatter launch's time a matter of where its buffers lie?  The same measurement in separate processes, device memory padded in front by PAD_MB
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
for pad in 0 0 0 1 3 17 64.5 129 333 1000.1 0; do
  echo "== PAD_MB=$pad"; PAD_MB=$pad IDENTICAL=0 SKIP_CHECK=1 PATHS=75 LAUNCH_FRAMES=12 REPS=40 timeout 300 python tools/ahead_time.py 2>&1 | grep -E "consecutive"
done
rocm-smi --showclocks 2>/dev/null | grep -E "sclk|mclk" | head -4
