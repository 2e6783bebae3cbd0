#!/usr/bin/env python3
"""Which frames should a rank of an 8-GPU run take?  (measurable on ONE GPU: every rank has its own GPU, its
own PCIe link and the whole cloud, so a rank's share rendered here alone is what that rank would do on a node.)

For a 128-frame (BASELINE configs[2]) and a 75-frame (the product's) video of the bench cloud, rank r's share as
  round-robin   cams[r::8]                           (the default, before and after this measurement)
  block         a contiguous run of n / 8 frames
  dealt<k>      runs of k consecutive frames dealt to the ranks in turn     (sharding.shard_indices)
delivered to pinned host memory and left in HBM, against the whole video on one GPU.  A launch group of consecutive cameras shares
its candidate lists (kbe_fused.hip: ShareMode): cameras 8 steps apart make them as wide as they get.  Prints one line per case:
frames, us per frame (median of PASSES passes), frames/s."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import common, sharding, synthetic  # noqa: E402

size = int(os.environ.get('SIZE', '1024'))
passes = int(os.environ.get('PASSES', '30'))
world = int(os.environ.get('WORLD', '8'))
dev = torch.device('cuda:0')
ofrom, oto = synthetic.default_windows(size, size, False)
base = {'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': False, 'dolly': False}
oc = bench.build_scene(size, dev, True, dict(base, dblSteps=np.linspace(0, 1, 75).tolist()), 1)
crop = common.crop_size(base)
host = torch.zeros(128, size, size, 3, dtype=torch.uint8, pin_memory=True)
devout = torch.zeros(128, size, size, 3, dtype=torch.uint8, device=dev)


def rate(cams, to_host):
    out = (host if to_host else devout)[:len(cams)]
    kw = dict(host_out=out) if to_host else dict(keep_on_device=True, host_out=out)
    for _ in range(3):
        common.render_frames(cams, oc, crop, **kw)
    ts = []
    for _ in range(passes):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        import time
        t0 = time.perf_counter()
        common.render_frames(cams, oc, crop, **kw)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts))


for steps in [int(v) for v in os.environ.get('VIDEOS', '128,75').split(',')]:
    path = common.frame_cameras(dict(base, dblSteps=np.linspace(0.0, 1.0, steps).tolist()), oc)
    for to_host in (True, False):
        t = rate(path, to_host)
        print('%3d-frame video, whole, %-9s: %3d frames %7.1f us per frame %8.0f frames/s' % (steps, 'delivered' if to_host else 'in HBM', steps, t / steps * 1e6, steps / t), flush=True)
    for shape in os.environ.get('SHAPES', 'round-robin,block,dealt4,dealt2').split(','):
        for to_host in (True, False):
            rs = []
            for r in (0, world // 2, world - 1):
                cams = [path[i] for i in sharding.shard_indices(steps, r, world, shape)]
                if not cams:
                    continue
                t = rate(cams, to_host)
                rs.append((r, len(cams), t))
            slowest = max(t for _, _, t in rs)
            print('%3d-frame video, %-11s share of %d ranks, %-9s: %s -> the video at the slowest rank\'s pace: %8.0f frames/s over %d GPUs (%.0f per rank)'
                  % (steps, shape, world, 'delivered' if to_host else 'in HBM',
                     '; '.join('rank %d: %d frames %6.1f us per frame' % (r, n, t / n * 1e6) for r, n, t in rs), steps / slowest, world, steps / slowest / world), flush=True)
