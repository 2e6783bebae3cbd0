#!/usr/bin/env python3
"""HIP-event times of the frame's launches (bench.time_kernels) for the bench scene -- with KBE_LIB_PATH a variant build
of the library (dev aid: what a change to a kernel does to the scatter alone on a stream)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

if os.environ.get('KBE_LIB_PATH'):
    _native._lib, _native._kernels, _native.LIB_PATH = None, None, os.environ['KBE_LIB_PATH']
size = int(os.environ.get('SIZE', '1024'))
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / 63 for i in range(64)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, torch.device('cuda:0'), os.environ.get('CLOUD', 'inpaint') == 'inpaint', settings, 1)
cams = common.frame_cameras(settings, oc)
only = os.environ.get('GROUP_ONLY')          # e.g. 4: only the launches with that many frames each (a clean rocprofv3 --stats of them)
if not only:
    kt = bench.time_kernels(oc, cams, 'fused')
    print(' '.join('%s=%.2f' % (k, v * 1e6) for k, v in kt.items() if k != 'route' and k.startswith(('bucket', 'fused'))))

# the bucket route's scatter (k_project + k_tiles) with 1..4 frames per launch, alone on a stream, us per FRAME
K = _native.kernels()
state = common._prepared_cloud(K, oc)
if True:
    focal, shift3 = cams[len(cams) // 2]
    out = torch.empty(4, size, size, 3, dtype=torch.uint8, device='cuda')
    for n in ((int(only),) if only else (1, 2, 3, 4)):
        group = [(focal, shift3)] * n

        def run(flag):
            K.render_frame_group(state, group, oc['dblBaseline'], out[:n], stages=3, zbuf_flags=[flag] * n)
        run(128); run(256)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for r in range(40):
            run(128 if r % 2 == 0 else 256)
        e1.record()
        torch.cuda.synchronize()
        print('scatter with %d frame(s) per launch: %.2f us per frame' % (n, e0.elapsed_time(e1) * 1e3 / 40 / n))

# the fused route's scatter (k_place + k_frame) with 1..4 frames per launch, alone on a stream, us per FRAME
K._pack(state)
focal, shift3 = cams[len(cams) // 2]
out = torch.empty(12, size, size, 3, dtype=torch.uint8, device='cuda')
K.group_scratch(state, 12)
for n in (1, 2, 4, 8, 12):
    group = [(focal, shift3)] * n
    par = [0]

    def run():
        K.render_frame_group_fused(state, group, oc['dblBaseline'], out[:n], stages=2, parities=[par[0] & 1] * n)
        par[0] += 1
    run(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(40):
        run()
    e1.record()
    torch.cuda.synchronize()
    print('fused scatter with %d frame(s) per launch: %.2f us per frame' % (n, e0.elapsed_time(e1) * 1e3 / 40 / n))
K.render_frame_group_fused(state, [(focal, shift3)] * 12, oc['dblBaseline'], out, stages=6)     # parity -1: leaves the sets' counters zeroed
