#!/usr/bin/env python3
"""us per frame of the bench workload, frames left in HBM or (HOST=1) delivered to pinned host memory; BATCH > 0 selects
the staged-ring hand-off (dev aid; KBE_LIB_PATH = variant build)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

if os.environ.get('KBE_LIB_PATH'):
    _native._lib, _native._kernels, _native.LIB_PATH = None, None, os.environ['KBE_LIB_PATH']

size = int(os.environ.get('SIZE', '1024'))
n = int(os.environ.get("FRAMES", "512"))
dolly = os.environ.get('DOLLY', '0') == '1'
ofrom, oto = synthetic.default_windows(size, size, dolly)
settings = {'dblSteps': [i / (n - 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': dolly}
dev = torch.device('cuda:0')
oc = bench.build_scene(size, dev, os.environ.get('CLOUD', 'inpaint') == 'inpaint' and not dolly and int(os.environ.get('UPSAMPLE', '1')) == 1, settings,
                       int(os.environ.get('UPSAMPLE', '1')))
cams = common.frame_cameras(settings, oc)
crop = common.crop_size(settings)
host = os.environ.get('HOST', '0') == '1'
batch = int(os.environ.get('BATCH', '0')) or None
out = torch.zeros(n, size, size, 3, dtype=torch.uint8, pin_memory=True) if host else torch.empty(n, size, size, 3, dtype=torch.uint8, device=dev)
kw = dict(host_out=out, batch=batch) if host else dict(keep_on_device=True, host_out=out)
common.render_frames(cams, oc, crop, **kw)
best = 1e9
enq = 1e9
for _ in range(int(os.environ.get("REPS", "7"))):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    common.render_frames(cams, oc, crop, **kw)
    enq = min(enq, (time.perf_counter() - t0) / n)
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / n)
print(('host-delivered' if host else 'device-only') + ' throughput: %.1f us/frame (host enqueue %.1f us/frame; %d frames, %d lanes, best of runs)' % (best * 1e6, enq * 1e6, n, int(os.environ.get('KBE_LANES', _native.DEFAULT_LANES))))
