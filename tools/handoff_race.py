#!/usr/bin/env python3
"""Does a delivered video ever differ from the same video left in HBM?  (dev aid, round 5: tools/video_soak.py found a few wrong bytes in one of
the LAST frames of two delivered videos in 1 200.)  One scene, PASSES passes of a FRAMES-frame video delivered to pinned host memory, each compared
with the frames left in HBM (a count of difference is the order of the fp32 sums: other launch groups; more is a wrong byte); the host reads the buffer right behind the stream's synchronisation.  KBE_LIB_PATH: a variant build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ken_burns_effect_amd import common, synthetic  # noqa: E402

H, W, n, passes = int(os.environ.get('H', '336')), int(os.environ.get('W', '456')), int(os.environ.get('FRAMES', '69')), int(os.environ.get('PASSES', '400'))
K = common._K()
image, disp = synthetic.make_rgbd(H, W, 5, 'noise')
depth = (512.0 * 120) / (disp + 1e-7)
oc = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H, 'objectDepthrange': synthetic.depthrange_of(depth),
      'tensorRawImage': image.cuda(), 'tensorRawDisparity': disp.cuda(), 'tensorRawDepth': depth.cuda()}
oc['tensorRawPoints'] = K.depth_to_points(oc['tensorRawDepth'], 512.0).view(1, 3, -1)
common._reset_inpa(oc)
ofrom, oto = synthetic.default_windows(H, W, False)
settings = {'dblSteps': [i / max(n - 1, 1) for i in range(n)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': False, 'dolly': False}
cams = common.frame_cameras(settings, oc)
want = common.render_frames(cams, oc, None, keep_on_device=True).cpu().numpy()
host = torch.zeros(n, H, W, 3, dtype=torch.uint8, pin_memory=True)
bad = 0
for p in range(passes):
    host.fill_(7)
    got = common.render_frames(cams, oc, None, host_out=host)
    d = np.abs(got.astype(np.int16) - want.astype(np.int16)) > 1
    if d.any():
        bad += 1
        fr = sorted(set(np.nonzero(d.reshape(n, -1).any(1))[0].tolist()))
        idx = np.nonzero(d.reshape(-1))[0]
        print('pass %d: %d bytes differ by more than a count, frames %s, byte offsets in the buffer %s (of %d), delivered %s, in HBM %s' % (p, int(d.sum()), fr, idx[:12].tolist(), d.size, got.reshape(-1)[idx[:12]].tolist(), want.reshape(-1)[idx[:12]].tolist()), flush=True)
print('%d of %d passes delivered other bytes than the frames left in HBM' % (bad, passes))
