#!/usr/bin/env python3
"""The device-side timeline of kbe_render_video when the frames go to pinned host memory: builds a variant library with
-DKBE_VIDEO_GPU_TRACE under /tmp (per transfer group, from events in the lanes' streams: when its rendering begins and ends, when its gate opens, when its transfer ends) and
renders FRAMES frames of the bench workload twice (dev aid)."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = '/tmp/libkbe_gpu_trace.so'
subprocess.check_call(['make', '-s', '-B', '-C', os.path.join(ROOT, 'ken-burns-effect_amd', 'csrc'), 'EXTRA=-DKBE_VIDEO_GPU_TRACE', 'OUT=' + so])
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

_native._lib, _native._kernels, _native.LIB_PATH = None, None, so
size, n = int(os.environ.get('SIZE', '1024')), int(os.environ.get('FRAMES', '20'))
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / (n - 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, torch.device('cuda:0'), True, settings, 1)
cams = common.frame_cameras(settings, oc)
crop = common.crop_size(settings)
out = torch.zeros(n, size, size, 3, dtype=torch.uint8, pin_memory=True)
for rep in range(3):
    torch.cuda.synchronize()
    sys.stderr.write('--- pass %d\n' % rep)
    t0 = time.perf_counter()
    common.render_frames(cams, oc, crop, host_out=out)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    sys.stderr.write('call returned after %.1f us, all frames landed after %.1f us (%.0f frames/s)\n' % ((t1 - t0) * 1e6, (t2 - t0) * 1e6, n / (t2 - t0)))
