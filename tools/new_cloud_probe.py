#!/usr/bin/env python3
"""What a video's FIRST render_frames on a new cloud costs beyond the frames themselves (dev aid): prepare_cloud + pack (sort),
scratch sets, the delivery probe -- every Pipeline call pays them, its cloud being new."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

size = int(os.environ.get('SIZE', '512'))
n = int(os.environ.get('FRAMES', '64'))
dev = torch.device('cuda:0')
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / (n - 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, dev, True, settings)
cams = common.frame_cameras(settings, oc)
crop = common.crop_size(settings)
host = torch.zeros(n, size, size, 3, dtype=torch.uint8, pin_memory=True)
K = _native.kernels()


def fresh():
    for k in ('tensorInpaPoints', 'tensorInpaImage', 'tensorInpaDepth'):
        oc[k] = oc[k].clone()
    oc.pop('_kbePreparedCloud', None)


def timed(fn, reps=8):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e6


common.render_frames(cams, oc, crop, host_out=host)
cached = timed(lambda: common.render_frames(cams, oc, crop, host_out=host))


def new_cloud():
    fresh()
    common.render_frames(cams, oc, crop, host_out=host)


first = timed(new_cloud)


def parts():
    fresh()
    t = [time.perf_counter()]
    state = common._prepared_cloud(K, oc); torch.cuda.synchronize(); t.append(time.perf_counter())
    K._pack(state); torch.cuda.synchronize(); t.append(time.perf_counter())
    K.delivery_lanes(state, cams, oc['dblBaseline'], crop); torch.cuda.synchronize(); t.append(time.perf_counter())
    K.render_video(state, cams, oc['dblBaseline'], crop, host_out=host); torch.cuda.synchronize(); t.append(time.perf_counter())
    K.render_video(state, cams, oc['dblBaseline'], crop, host_out=host); torch.cuda.synchronize(); t.append(time.perf_counter())
    return [(b - a) * 1e6 for a, b in zip(t[:-1], t[1:])]


ps = [parts() for _ in range(6)]
med = [sorted(p[i] for p in ps)[len(ps) // 2] for i in range(5)]
print('%d^2, %d frames, %d points: render_frames on a cached cloud %.0f us, on a new cloud %.0f us; parts (each synchronised): prepare_cloud %.0f, pack %.0f, '
      'delivery probe %.0f, first render_video %.0f, second %.0f' % (size, n, oc['tensorInpaPoints'].shape[-1], cached, first, *med))
