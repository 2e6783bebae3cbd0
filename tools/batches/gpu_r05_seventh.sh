#!/bin/bash
# GPU box (round 5): spilled records as point indices (4 B instead of 16 B, four times the capacity), candidate lists of 1024 entries: the density
# sweep, configs[4], the headline workload, the GPU suite
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_seventh
mkdir -p $O
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
r=d['roofline']
print('%.0f delivered, %.0f left in HBM (%.1f us per frame), ok %s; roofline %.2f us per frame -> %.4f' % (d['value'] or -1, d['device_only']['value'], d['device_only']['ms_per_step']*1e3, d['frames_check']['ok'], r['us_per_frame'], r['frac']))"; }
for lib in "" $R/_variants/cand1024.so; do
  echo "== library: ${lib:-shipped}"
  echo "config4: $(KBE_LIB_PATH=$lib timeout 900 python bench.py --no-cpu-baseline --size 2048 --upsample 2 --steps 64 --warmup 8 2>/dev/null | val)"
  echo "default --steps 75: $(KBE_LIB_PATH=$lib timeout 900 python bench.py --no-cpu-baseline --steps 75 --warmup 20 2>/dev/null | val)"
  KBE_LIB_PATH=$lib ROUTES=1,0 timeout 1200 python tools/density_sweep.py 2>&1 | tee $O/density_$(basename ${lib:-shipped}).txt | grep "points per pixel"
done
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pd_$c
  SIZE=2048 UPSAMPLE=2 CLOUD=raw KBE_LANES=1 FRAMES=16 REPS=1 timeout 400 rocprofv3 --pmc $c -d /tmp/pd_$c -o c --output-format csv -- python $R/tools/throughput.py > /tmp/pd.log 2>&1 || tail -3 /tmp/pd.log
  python $R/tools/pmc_by_grid.py /tmp/pd_$c/c_counter_collection.csv k_frame_group_ahead k_frame_group k_place 2>&1 | cut -c1-300
done | tee $O/config4_traffic.txt
cd $R; echo "== gpu tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
