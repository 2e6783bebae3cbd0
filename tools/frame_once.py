#!/usr/bin/env python3
"""Renders a handful of frames of the bench workload -- the command rocprofv3 wraps (dev aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

if os.environ.get('KBE_LIB_PATH'):          # a variant build (tools/gpu_variant_pmc.sh)
    _native._lib, _native._kernels, _native.LIB_PATH = None, None, os.environ['KBE_LIB_PATH']

size = int(os.environ.get('SIZE', '1024'))
n = int(os.environ.get('FRAMES', '9'))
dolly = os.environ.get('DOLLY', '0') == '1'
ofrom, oto = synthetic.default_windows(size, size, dolly)
settings = {'dblSteps': [i / (n - 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': dolly}
oc = bench.build_scene(size, torch.device('cuda:0'), os.environ.get('CLOUD', 'inpaint') == 'inpaint' and not dolly and int(os.environ.get('UPSAMPLE', '1')) == 1, settings,
                       int(os.environ.get('UPSAMPLE', '1')))
import time  # noqa: E402
for _ in range(int(os.environ.get('REPS', '1'))):       # REPS > 1: bursts 20 ms apart (tools/handoff_timeline.py shows the last one)
    frames = common.render_frames(common.frame_cameras(settings, oc), oc, common.crop_size(settings), overlap=False)
    time.sleep(0.02)
print(frames.shape, oc['tensorInpaPoints'].shape[-1])
