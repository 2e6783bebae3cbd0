export HSA_ENABLE_IPC_MODE_LEGACY=0
R=$GRAFT_REPO_ROOT
cd /tmp
for n in 6 8 12 20 24 30 40; do
  for v in tail0 default tail0 default; do
    envs="KBE_LIB_PATH=$R/_variants/$v.so"; [ $v = default ] && envs="KBE_NONE=1"
    echo "$n frames, $v: $(env $envs timeout 300 python $R/bench.py --no-cpu-baseline --device-only --steps $n --warmup 8 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"]), d["config"]["pass_ms"])')"
  done
done
