export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', d['config']['pass_ms'], d['frames_check']['ok'])"; }
for spec in "75:0 -24 -32 0 -24 -32" "40:0 -16 -20 0 -16" "12:0 -8 0 -8" "30:0 -8 -16 0"; do
steps=${spec%%:*}
for b in ${spec#*:}; do
  echo "== --steps $steps KBE_DELIVERY_BATCH=$b"; KBE_DELIVERY_BATCH=$b timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 5 2>/dev/null | line
done
done
