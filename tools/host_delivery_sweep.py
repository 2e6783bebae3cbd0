#!/usr/bin/env python3
"""ms for N frames delivered to pinned host memory, per (lanes, batch): min / median of several runs (dev aid)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import common, synthetic  # noqa: E402

size, n = 1024, int(os.environ.get('FRAMES', '75'))
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / (n - 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
dev = torch.device('cuda:0')
oc = bench.build_scene(size, dev, True, settings)
cams = common.frame_cameras(settings, oc)
crop = common.crop_size(settings)
host = torch.empty(n, size, size, 3, dtype=torch.uint8, pin_memory=True)
for lanes in (1, 2, 3, 4):
    for batch in [int(b) for b in os.environ.get('BATCHES', '8,19,38,75').split(',')]:
        os.environ['KBE_LANES'] = str(lanes)
        oc.pop('_kbePreparedCloud', None)
        common.render_frames(cams, oc, crop, host_out=host, batch=batch)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            common.render_frames(cams, oc, crop, host_out=host, batch=batch)
            ts.append((time.perf_counter() - t0) * 1e3)
        print('lanes %d batch %2d: min %.2f ms  median %.2f ms  (%.0f frames/s best)' % (lanes, batch, min(ts), float(np.median(ts)), n / min(ts) * 1e3), flush=True)
