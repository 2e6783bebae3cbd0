#!/bin/bash
# GPU box (round 4): the whole GPU suite, then bench.py at the driver's and the product's video lengths with both ramps of the transfer groups
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R && timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee $O/c_tests.txt
for ramp in classic fast classic fast; do
  for steps in 20 75; do
    KBE_RAMP=$ramp timeout 600 python bench.py --steps $steps --warmup 20 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
    python - <<P
import json
d = json.loads(open('/tmp/line.json').read())
print('ramp $ramp steps $steps: %.0f frames/s delivered (%.3f ms per pass of $steps), device-only %.0f, pcie %.1f GB/s, passes %d, check %s' % (d['value'], d['ms_per_step'] * $steps, d['device_only']['value'], d['pcie']['achieved'], d['config']['passes'], d['frames_check']['ok']))
P
  done
done 2>&1 | tee $O/c_short.txt
KBE_RAMP=fast timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/c_bench_1024steps_fast.json
KBE_RAMP=classic timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/c_bench_1024steps_classic.json
python - <<P
import json
for r in ('fast', 'classic'):
    d = json.loads(open('$O/c_bench_1024steps_%s.json' % r).read())
    print(r, 'steps 1024: %.0f delivered, %.0f device-only, roofline %.3f' % (d['value'], d['device_only']['value'], d['roofline']['frac']))
P
