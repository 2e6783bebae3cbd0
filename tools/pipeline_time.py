#!/usr/bin/env python3
"""BASELINE.json configs[1]: one image through the whole Pipeline (estimation + refinement + inpainting nets with
seeded weights, then N frames): wall time of the second call (dev aid)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ken_burns_effect_amd import kbe, synthetic
from ken_burns_effect_amd.pipeline import Pipeline
size, n = int(os.environ.get('SIZE', '512')), int(os.environ.get('FRAMES', '64'))
image, _ = synthetic.make_rgbd(size, size, 9)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    pipe = Pipeline(model_paths=None, allow_random_weights=True, device='cuda:0', steps=n)
zoom = kbe.windows_for(size, size, dict.fromkeys(('startU', 'startV', 'startW', 'startH', 'endU', 'endV', 'endW', 'endH')), False)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    frames = pipe(image, zoom)
    torch.cuda.synchronize()
    print('call %d: %.1f ms for %d frames of %dx%d' % (i, (time.perf_counter() - t0) * 1e3, len(frames), size, size), flush=True)
