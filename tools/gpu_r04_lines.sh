#!/bin/bash
export HSA_ENABLE_IPC_MODE_LEGACY=0 MIOPEN_FIND_MODE=FAST
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r04
mkdir -p $O
cd $R
timeout 900 python bench.py --no-cpu-baseline --dolly --steps 128 --warmup 16 2> $O/g_dolly.err | tail -1 > $O/g_bench_dolly.json; tail -3 $O/g_dolly.err
timeout 900 python bench.py --no-cpu-baseline --size 2048 --upsample 2 --steps 32 --warmup 8 2> $O/g_c4.err | tail -1 > $O/g_bench_config4.json; tail -3 $O/g_c4.err
timeout 900 python bench.py --no-cpu-baseline --size 512 --steps 512 --warmup 64 2> $O/g_512.err | tail -1 > $O/g_bench_512.json; tail -3 $O/g_512.err
python - <<P
import json
for n in ('dolly', 'config4', '512'):
    try:
        d = json.loads(open('$O/g_bench_%s.json' % n).read())
        r = d['roofline']
        print(n, '%.0f delivered, %.0f device-only (%.1f us/frame)' % (d['value'], d['device_only']['value'], 1e6 / d['device_only']['value']), '| roofline:', r['kernel'], 'frac %.3f us/frame %s' % (r['frac'], r.get('us_per_frame')),
              '|', {k: (round(v['frac'], 3), v.get('us_per_frame')) for k, v in r.items() if isinstance(v, dict) and 'frac' in v})
    except Exception as e:
        print(n, 'FAILED', e)
P
