#!/bin/bash
# GPU box: delivered frames/s of short videos (FRAMES=20, 75) for hand-off schedules (dev aid)
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
for n in 20 75; do
for cfg in "" "KBE_EVEN_GROUPS=1" "KBE_DELIVERY_BATCH=-3" "KBE_DELIVERY_BATCH=-3 KBE_EVEN_GROUPS=1" "KBE_DELIVERY_BATCH=-4 KBE_EVEN_GROUPS=1" "KBE_DELIVERY_BATCH=-8" "KBE_DELIVERY_BATCH=-8 KBE_HOST_LANES=3" "KBE_DELIVERY_BATCH=-4 KBE_HOST_LANES=3" "KBE_DELIVERY_BATCH=-4 KBE_HOST_LANES=3 KBE_EVEN_GROUPS=1" "KBE_DELIVERY_BATCH=-16 KBE_HOST_LANES=4"; do
  echo "== n=$n $cfg: $(env HOST=1 FRAMES=$n REPS=15 $cfg timeout 300 python $R/tools/throughput.py 2>/dev/null | tail -1 | cut -c1-60)"
done
done
