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
kt = bench.time_kernels(oc, cams)
print(' '.join('%s=%.2f' % (k, v * 1e6) for k, v in kt.items() if k != 'route' and k.startswith(('bucket', 'fused'))))
