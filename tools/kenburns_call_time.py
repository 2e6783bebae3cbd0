#!/usr/bin/env python3
"""Wall time of common.process_kenburns (frame loop only, boolInpaint False) as a user calls it (dev aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ken_burns_effect_amd import common, synthetic
size, n = 1024, int(os.environ.get('FRAMES', '75'))
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / (n - 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': False, 'dolly': False}
oc = bench.build_scene(size, torch.device('cuda:0'), True, dict(settings, boolInpaint=True))
for i in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frames = common.process_kenburns(settings, oc, None)
    dt = time.perf_counter() - t0
    print('call %d: %.1f ms for %d frames (%s %s)' % (i, dt * 1e3, len(frames), frames[0].shape, frames[0].dtype), flush=True)
