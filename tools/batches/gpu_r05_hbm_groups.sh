export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
cd ${GRAFT_REPO_ROOT:-/root/repo}
line() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value'],1), 'delivered', round(d['device_only']['value'],1), 'left in HBM', d['frames_check']['ok'], 'lanes', d['config']['lanes'])"; }
for steps in 256 75; do
for g in 4 6 8 12 4 6; do
  echo "== steps $steps KBE_FILL_GROUP=$g"; KBE_FILL_GROUP=$g timeout 300 python bench.py --no-cpu-baseline --steps $steps --warmup 32 2>/dev/null | line
done
done
