#!/usr/bin/env python3
"""What the distance-table hole fill does per frame of a dolly zoom (dev aid): holes walked, table look-ups per hole,
directions cut by the bound.  Builds a -DKBE_FRAME_STATS variant library under /tmp, like tools/frame_stats.py."""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = '/tmp/libkbe_stats.so'
src = os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc')
subprocess.check_call(['make', '-s', '-B', '-C', src, 'EXTRA=-DKBE_FRAME_STATS ' + os.environ.get('EXTRA', ''), 'OUT=' + so])
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

_native._lib, _native._kernels, _native.LIB_PATH = None, None, so
size = int(os.environ.get('SIZE', '1024'))
ofrom, oto = synthetic.default_windows(size, size, True)
settings = {'dblSteps': [i / 8 for i in range(9)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': True}
oc = bench.build_scene(size, torch.device('cuda:0'), False, settings, 1)
K = _native.kernels()
state = common._prepared_cloud(K, oc)
out = (ctypes.c_ulonglong * 8)()
hist = (ctypes.c_ulonglong * 16)()
ex = torch.empty(size * size, device='cuda')
for focal, shift3 in common.frame_cameras(settings, oc):
    torch.cuda.synchronize()
    K.lib.kbe_debug_fill_stats(out, 1)
    K.lib.kbe_debug_fill_hist(hist, 1)
    K.render_frame(state, shift3, focal, oc['dblBaseline'], existing_f32=ex, stages=7 | 8 | 512, fused=False)
    torch.cuda.synchronize()
    K.lib.kbe_debug_fill_stats(out, 0)
    holes = int((ex <= 0).sum())
    K.lib.kbe_debug_fill_hist(hist, 0)
    print('   ray ends by loop iterations lived (< 4, < 8, < 16, ...): %s; the longest %d; rays of >= 128 iterations take %.2f steps per iteration'
          % (' '.join(str(hist[i]) for i in range(12)), hist[12], hist[13] / max(1, hist[14])))
    hit_wait, out[0] = out[0] >> 32, out[0] & 0xFFFFFFFF
    h = max(1, out[0])
    it, it_drain = out[6] & 0xFFFFFFFF, out[6] >> 32
    walking, walking_drain = out[7] & 0xFFFFFFFF, out[7] >> 32
    print('holes %d, inside the box of valid pixels %d; per hole: directions walked %.1f (cut by the bound %.1f); fine look-ups %.1f, coarse look-ups %.1f; '
          'catch-ups %.2f per look-up; wave iterations %d (%d with the queue empty), lanes walking per iteration %.1f of 64 (%.1f while the queue has work, %.1f after), waiting for the other end %.1f'
          % (holes, out[0], out[1] / h, out[5] / h, out[2] / h, out[3] / h, out[4] / max(1, out[3]), it, it_drain, walking / max(1, it),
             (walking - walking_drain) / max(1, it - it_drain), walking_drain / max(1, it_drain), hit_wait / max(1, it)))
