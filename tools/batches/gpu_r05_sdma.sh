#!/bin/bash
# GPU box (round 5): the frame hand-off by an SDMA engine (KBE_HANDOFF=sdma -> KBE_VIDEO_SDMA) against hipMemcpyAsync (blit), driver-shaped runs
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_sdma
mkdir -p $O
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.0f delivered (%.3f ms per pass, lanes %s, pcie %.1f GB/s), %.0f left in HBM, ok %s' % (d['value'] or -1, d['config']['pass_ms']['median'], d['config']['lanes'], d['pcie']['achieved'], d['device_only']['value'], d['frames_check']))"; }
for rep in 1 2; do
for h in blit sdma; do
  for args in "--steps 20 --warmup 5" "--steps 75 --warmup 20" "--steps 1024 --warmup 128"; do
    echo "$h [$args]: $(KBE_HANDOFF=$h timeout 600 python bench.py --no-cpu-baseline $args 2>$O/err_$h.txt | tee $O/bench_${h}_$(echo $args | tr -d ' -').json | val)"
  done
done
done
for l in 1 2 3 4; do
  echo "sdma lanes $l [--steps 20]: $(KBE_HANDOFF=sdma KBE_HOST_LANES=$l timeout 600 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | val)"
  echo "sdma lanes $l [--steps 75]: $(KBE_HANDOFF=sdma KBE_HOST_LANES=$l timeout 600 python bench.py --no-cpu-baseline --steps 75 --warmup 20 2>/dev/null | val)"
done
tail -3 $O/err_sdma.txt
