#!/usr/bin/env python3
"""The tile counters of a delivered video (dev aid): a -DKBE_FRAME_STATS [-DKBE_SHARED_LISTS=1] build renders FRAMES frames of a
cloud to pinned host memory and reports list entries, candidates, wide sub-blocks and slow tiles per frame."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = '/tmp/libkbe_share_stats.so'
flags = '-DKBE_FRAME_STATS' + (' -DKBE_SHARED_LISTS=1' if os.environ.get('SHARE', '1') == '1' else '')
subprocess.check_call(['make', '-s', '-B', '-C', os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc'), 'EXTRA=' + flags, 'OUT=' + so])
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

_native._lib, _native._kernels, _native.LIB_PATH = None, None, so
size, up, n = int(os.environ.get('SIZE', '2048')), int(os.environ.get('UPSAMPLE', '2')), int(os.environ.get('FRAMES', '64'))
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / (n - 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, torch.device('cuda:0'), up == 1, settings, up)
cams = common.frame_cameras(settings, oc)
crop = common.crop_size(settings)
host = torch.zeros(n, size, size, 3, dtype=torch.uint8, pin_memory=True)
K = _native.kernels()
out = (ctypes.c_ulonglong * 8)()
for rep in range(2):
    torch.cuda.synchronize()
    K.lib.kbe_debug_frame_stats(out, 1)
    common.render_frames(cams, oc, crop, host_out=host)
    torch.cuda.synchronize()
    K.lib.kbe_debug_frame_stats(out, 1)
    t = max(1, out[0])
    print('%s: tiles %d: list entries %.1f, candidate sub-blocks %.1f, points in z reach %.0f, records %.0f per tile; wide sub-blocks %d; tiles on the slow path %d, with a second round %d'
          % (flags, out[0], out[1] / t, out[2] / t, out[3] / t, out[4] / t, out[7], out[5], out[6]))
