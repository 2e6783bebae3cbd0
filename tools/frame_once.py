#!/usr/bin/env python3
"""Renders a handful of frames at SIZE (default 1024) -- the command rocprofv3 wraps (dev aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

size = int(os.environ.get('SIZE', '1024'))
n = int(os.environ.get('FRAMES', '9'))
oc = bench.build_scene(size, torch.device('cuda:0'), False)
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / (n - 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': False, 'dolly': False}
frames = common.render_frames(common.frame_cameras(settings, oc), oc, common.crop_size(settings))
print(frames.shape)
