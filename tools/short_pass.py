#!/usr/bin/env python3
"""Renders a FRAMES-frame delivered video of the bench cloud PASSES times, a host synchronisation between the passes (dev aid: the
workload tools/batches/gpu_r05_timeline.sh traces; tools/sdma_timeline.py lays the last pass out)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import common, synthetic  # noqa: E402

size, n, passes = int(os.environ.get('SIZE', '1024')), int(os.environ.get('FRAMES', '20')), int(os.environ.get('PASSES', '8'))
dev = torch.device('cuda:0')
ofrom, oto = synthetic.default_windows(size, size, False)
base = {'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, dev, True, dict(base, dblSteps=[0.0, 1.0]))
settings = dict(base, dblSteps=[i / (n - 1) for i in range(n)])
cams = common.frame_cameras(settings, oc)
crop = common.crop_size(settings)
host = torch.zeros(n, size, size, 3, dtype=torch.uint8, pin_memory=True)
for _ in range(passes):
    torch.cuda.synchronize()
    time.sleep(0.002)
    t0 = time.perf_counter()
    common.render_frames(cams, oc, crop, host_out=host)
    print('pass of %d frames: %.0f us' % (n, (time.perf_counter() - t0) * 1e6), flush=True)
