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


def host_parts():
    """the same without synchronising in between: where the HOST spends the call"""
    fresh()
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    state = common._prepared_cloud(K, oc); t.append(time.perf_counter())
    K._pack(state); t.append(time.perf_counter())
    flags, group, fused = K.video_launch_shape(state, cams, None, to_host=True); t.append(time.perf_counter())
    K.group_scratch(state, group * 2); t.append(time.perf_counter())
    K.render_video(state, cams, oc['dblBaseline'], crop, host_out=host); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    return [(b - a) * 1e6 for a, b in zip(t[:-1], t[1:])]


hs = [host_parts() for _ in range(6)]
hmed = [sorted(p[i] for p in hs)[len(hs) // 2] for i in range(6)]
print('host time of a new cloud\'s call, unsynchronised: prepare_cloud %.0f us, pack %.0f, launch shape %.0f, scratch sets %.0f, render_video %.0f, wait for the GPU %.0f' % tuple(hmed))
ps = [parts() for _ in range(6)]
med = [sorted(p[i] for p in ps)[len(ps) // 2] for i in range(5)]
print('%d^2, %d frames, %d points: render_frames on a cached cloud %.0f us, on a new cloud %.0f us; parts (each synchronised): prepare_cloud %.0f, pack %.0f, '
      'delivery probe %.0f, first render_video %.0f, second %.0f' % (size, n, oc['tensorInpaPoints'].shape[-1], cached, first, *med))

if os.environ.get('PROFILE') == '1':
    import cProfile
    import pstats
    pr = cProfile.Profile()
    for _ in range(5):
        fresh()
        torch.cuda.synchronize()
        pr.enable()
        common.render_frames(cams, oc, crop, host_out=host)
        torch.cuda.synchronize()
        pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)

if os.environ.get('FRESH') == '1':
    import gc
    for _ in range(4):
        common.render_frames(cams, oc, crop, host_out=host)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in ('tensorInpaPoints', 'tensorInpaImage', 'tensorInpaDepth'):
            oc[k] = oc[k].clone()
        t1 = time.perf_counter()
        old = oc.pop('_kbePreparedCloud', None)
        t2 = time.perf_counter()
        keys = sorted(old[1].keys())
        sizes = {k: (v.numel() * v.element_size() >> 20) for k, v in old[1].items() if torch.is_tensor(v)}
        del old
        t3 = time.perf_counter()
        torch.cuda.synchronize()
        t4 = time.perf_counter()
        print('fresh(): clones %.0f us, pop %.0f, dropping the state %.0f, synchronize %.0f; state tensors (MB): %s' % ((t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6, sizes))
