#!/usr/bin/env python3
"""BASELINE configs[1] as a whole (a 512^2 image through Pipeline: three networks, two inpaint passes, 64 delivered frames) for
rocprofv3 --kernel-trace: two warm calls, a marker launch (k_selftest_division), then CALLS timed calls.  With a trace CSV as the
argument it reports instead: GPU time per call by kernel after the LAST marker, largest first, and by kind (dev aid).
    cd /tmp && rocprofv3 --kernel-trace -d out -o t --output-format csv -- python $REPO/tools/pipeline_trace.py
    python $REPO/tools/pipeline_trace.py out/t_kernel_trace.csv"""
import collections
import csv
import os
import sys

CALLS = int(os.environ.get('CALLS', '5'))
if len(sys.argv) > 1:
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    last = max(i for i, r in enumerate(rows) if 'k_selftest_division' in r['Kernel_Name'])
    per, calls = collections.defaultdict(float), collections.Counter()
    for r in rows[last + 1:]:
        name = r['Kernel_Name']
        per[name] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 / CALLS
        calls[name] += 1

    def kind(n):
        if '(anonymous namespace)::k_' in n[:40] or 'kbe::' in n:
            return 'this library'
        if 'copyBuffer' in n or 'fillBuffer' in n:
            return 'runtime copies / fills'
        if any(k in n for k in ('miopen', 'igemm', 'Cijk_', 'Im2d2Col', 'batched_transpose', 'SubTensorOp', 'naive_conv', 'ck16tensor', 'ck5', 'gridwise', 'Winograd', 'Sp3AsmConv', 'conv')):
            return 'MIOpen convolutions (+ their transposes, bias adds)'
        return 'PyTorch element-wise / pooling / resampling'
    kinds = collections.defaultdict(float)
    for n, t in per.items():
        kinds[kind(n)] += t
    total = sum(per.values())
    print('GPU time per call: %.2f ms in %d launches (sum of kernel durations; kernels of different streams overlap)' % (total / 1e3, sum(calls.values()) // CALLS))
    for k, t in sorted(kinds.items(), key=lambda kv: -kv[1]):
        print('  %6.2f ms  %4.1f %%  %s' % (t / 1e3, 100 * t / total, k))
    only = os.environ.get('KIND')                # e.g. KIND=PyTorch: the launches of that kind only
    for n, t in sorted(((n, t) for n, t in per.items() if not only or kind(n).startswith(only)), key=lambda kv: -kv[1])[:int(os.environ.get('TOP', '40'))]:
        print('%8.1f us  %5d launches per call  avg %7.1f us  %s' % (t, calls[n] // CALLS, t * CALLS / calls[n], n[:150]))
    sys.exit(0)

sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch  # noqa: E402

from ken_burns_effect_amd import _native, kbe, synthetic  # noqa: E402
from ken_burns_effect_amd.pipeline import Pipeline  # noqa: E402

size = int(os.environ.get('SIZE', '512'))
image, _ = synthetic.make_rgbd(size, size, 9)
pipe = Pipeline(model_paths=None, allow_random_weights=True, device='cuda:0', steps=64)
zoom = kbe.windows_for(size, size, dict.fromkeys(('startU', 'startV', 'startW', 'startH', 'endU', 'endV', 'endW', 'endH')), False)
for _ in range(2):
    pipe(image, zoom)
torch.cuda.synchronize()
_native.kernels().selftest_division(torch.ones(64, device='cuda'), torch.ones(64, device='cuda'))
torch.cuda.synchronize()
for _ in range(CALLS):
    frames = pipe(image, zoom)
torch.cuda.synchronize()
print(len(frames), frames[0].shape)
