#!/usr/bin/env python3
"""Where a short delivered video's time goes (dev aid): per pass of FRAMES frames -- the host's share of render_frames (until the
call has enqueued everything), the wall time to the last byte, and a cProfile of the host side."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

size = int(os.environ.get('SIZE', '1024'))
dev = torch.device('cuda:0')
ofrom, oto = synthetic.default_windows(size, size, False)
base = {'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, dev, True, dict(base, dblSteps=[0.0, 1.0]))
K = _native.kernels()
for n in [int(v) for v in os.environ.get('FRAMES', '20,75').split(',')]:
    settings = dict(base, dblSteps=[i / (n - 1) for i in range(n)])
    cams = common.frame_cameras(settings, oc)
    crop = common.crop_size(settings)
    host = torch.zeros(n, size, size, 3, dtype=torch.uint8, pin_memory=True)
    for _ in range(5):
        common.render_frames(cams, oc, crop, host_out=host)
    state = common._prepared_cloud(K, oc)
    enq, wall = [], []
    for _ in range(30):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        K.render_video(state, cams, oc['dblBaseline'], crop, host_out=host)
        t1 = time.perf_counter()
        torch.cuda.current_stream().synchronize()
        t2 = time.perf_counter()
        enq.append(t1 - t0)
        wall.append(t2 - t0)
    enq.sort()
    wall.sort()
    print('%d frames: host enqueue median %.0f us (min %.0f), wall median %.0f us (min %.0f) = %.0f frames/s; link-only %.0f us' % (
        n, enq[15] * 1e6, enq[0] * 1e6, wall[15] * 1e6, wall[0] * 1e6, n / wall[15], n * 3 * size * size / 55e9 * 1e6), flush=True)
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        K.render_video(state, cams, oc['dblBaseline'], crop, host_out=host)
        torch.cuda.current_stream().synchronize()
    pr.disable()
    st = pstats.Stats(pr, stream=sys.stdout)
    st.sort_stats('cumulative').print_stats(14)
