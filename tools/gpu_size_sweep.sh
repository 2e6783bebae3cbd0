export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
for size in 384 640 768 896 1280 1536; do
  for cfg in "KBE_FUSED=0" "KBE_FUSED=1" "KBE_FUSED=1 KBE_FILL_GROUP=2" "KBE_FUSED=1 KBE_FILL_GROUP=4"; do
    echo "== SIZE=$size $cfg: $(env SIZE=$size $cfg FRAMES=128 REPS=3 timeout 300 python $R/tools/throughput.py 2>&1 | tail -1 | cut -c1-50)"
  done
done
