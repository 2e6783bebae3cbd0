#!/bin/bash
# GPU box (round 5): SDMA hand-off with two halves of slots per lane (lanes 2 / 3 / 4), the shapes of a rank's share on this tree,
# consecutive against strided frames per launch for frames left in HBM, and the GPU suite
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r05_third
mkdir -p $O
cd $R
val() { python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%.0f delivered (%.3f ms per pass, lanes %s, pcie %.1f GB/s), %.0f left in HBM, ok %s' % (d['value'] or -1, d['config']['pass_ms']['median'], d['config']['lanes'], d['pcie']['achieved'], d['device_only']['value'], d['frames_check']['ok']))"; }
for h in blit sdma; do
  for l in 2 3 4; do
    for args in "--steps 20 --warmup 5" "--steps 75 --warmup 20"; do
      echo "$h lanes $l [$args]: $(KBE_HANDOFF=$h KBE_HOST_LANES=$l timeout 600 python bench.py --no-cpu-baseline $args 2>/dev/null | val)"
    done
  done
done
echo "sdma [--steps 1024]: $(KBE_HANDOFF=sdma timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | val)"
for h in blit sdma; do echo "== shard shapes, $h"; KBE_HANDOFF=$h timeout 900 python tools/shard_shapes.py 2>&1 | tee $O/shard_shapes_$h.txt | grep -E "video"; done
echo "== frames left in HBM: strided"; for a in "--steps 75 --warmup 20" "--steps 128 --warmup 20"; do KBE_LIB_PATH=$R/_variants/strided.so timeout 600 python bench.py --no-cpu-baseline --device-only $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])"; done
echo "== frames left in HBM: consecutive"; for a in "--steps 75 --warmup 20" "--steps 128 --warmup 20"; do timeout 600 python bench.py --no-cpu-baseline --device-only $a 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'])"; done
echo "== strong-scaling line, one rank"; timeout 600 python bench.py --no-cpu-baseline --video-frames 128 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['value'], d['scaling'], d['steps'], d['config']['frames_per_rank'], d['roofline']['frac'], d['roofline']['cameras'])"
echo "== gpu tests"; timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -5
