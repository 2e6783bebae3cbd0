#!/usr/bin/env python3
"""Where does the HOST spend a short delivered video's call?  For a FRAMES-frame video of the bench cloud: the wall time of the pass, of the
call until it returns (everything is enqueued), and cProfile's view of the Python side of PASSES calls (dev aid, round 5)."""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

size, n, passes = int(os.environ.get('SIZE', '1024')), int(os.environ.get('FRAMES', '20')), int(os.environ.get('PASSES', '200'))
dev = torch.device('cuda:0')
ofrom, oto = synthetic.default_windows(size, size, False)
base = {'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, dev, True, dict(base, dblSteps=[0.0, 1.0]))
settings = dict(base, dblSteps=[i / (n - 1) for i in range(n)])
cams = common.frame_cameras(settings, oc)
crop = common.crop_size(settings)
host = torch.zeros(n, size, size, 3, dtype=torch.uint8, pin_memory=True)
K = _native.kernels()
state = common._prepared_cloud(K, oc)
for _ in range(10):
    common.render_frames(cams, oc, crop, host_out=host)
tot, enq = [], []
for _ in range(passes):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K.render_video(state, cams, oc['dblBaseline'], crop, host_out=host)
    t1 = time.perf_counter()
    torch.cuda.current_stream().synchronize()
    t2 = time.perf_counter()
    tot.append(t2 - t0); enq.append(t1 - t0)
print('%d-frame video: pass %.0f us (median of %d), of which the call until it returns %.0f us' % (n, np.median(tot) * 1e6, passes, np.median(enq) * 1e6))
# the C call alone: patch the library entry with a timer
lib = K.lib
inner = []
real = lib.kbe_render_video
def timed(*a):
    t = time.perf_counter(); r = real(*a); inner.append(time.perf_counter() - t); return r
class Shim:
    def __getattr__(self, k):
        return timed if k == 'kbe_render_video' else getattr(lib, k)
K.lib = Shim()
for _ in range(passes):
    torch.cuda.synchronize()
    K.render_video(state, cams, oc['dblBaseline'], crop, host_out=host)
torch.cuda.synchronize()
K.lib = lib
print('  the C call kbe_render_video alone: %.0f us (median)' % (np.median(inner) * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(passes):
    common.render_frames(cams, oc, crop, host_out=host)
pr.disable()
st = pstats.Stats(pr, stream=sys.stdout).sort_stats('cumulative')
st.print_stats(18)
