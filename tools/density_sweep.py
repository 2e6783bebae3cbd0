#!/usr/bin/env python3
"""Where do the tile routes fall off their cliff?  (dev aid)  A 1024 x 1024 raster from a 3 x 3-upsampled cloud
(9.4 M points), thinned at random to 5, 6, 7, 8 and 9 points per pixel; 16 frames left in HBM on the bucket route (KBE_FUSED=0),
the fused route (1) and the stage-by-stage atomic kernels (generic): us per frame."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

size, up = int(os.environ.get('SIZE', '1024')), int(os.environ.get('UP', '3'))
dev = torch.device('cuda:0')
K = _native.kernels()
image_u, disp_u = synthetic.make_rgbd(size * up, size * up, 0)
depth_u = ((synthetic.FOCAL * synthetic.BASELINE) / (disp_u + 1e-7)).to(dev)
pts = K.depth_to_points(depth_u, synthetic.FOCAL * up).view(1, 3, -1)
img = image_u.to(dev).reshape(1, 3, -1)
dep = depth_u.reshape(1, 1, -1)
n_all = pts.shape[2]
perm = torch.randperm(n_all, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
image, disp = synthetic.make_rgbd(size, size, 0)
depth = (synthetic.FOCAL * synthetic.BASELINE) / (disp + 1e-7)
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': np.linspace(0, 1, 16).tolist(), 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': False, 'dolly': False}
out = torch.empty(16, size, size, 3, dtype=torch.uint8, device=dev)
for density in [float(v) for v in os.environ.get('DENSITIES', '4,5,6,7,8,9').split(',')]:
    n = min(n_all, int(density * size * size))
    keep = perm[:n].sort().values
    oc = {'dblFocal': synthetic.FOCAL, 'dblBaseline': synthetic.BASELINE, 'intWidth': size, 'intHeight': size, 'objectDepthrange': synthetic.depthrange_of(depth),
          'tensorInpaPoints': pts[:, :, keep].contiguous(), 'tensorInpaImage': img[:, :, keep].contiguous(), 'tensorInpaDepth': dep[:, :, keep].contiguous()}
    cams = common.frame_cameras(settings, oc)
    res = []
    for route in os.environ.get('ROUTES', '0,1,generic').split(','):
        os.environ['KBE_FUSED'] = route
        oc.pop('_kbePreparedCloud', None)
        try:
            common.render_frames(cams, oc, None, keep_on_device=True, host_out=out)
            torch.cuda.synchronize()
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                common.render_frames(cams, oc, None, keep_on_device=True, host_out=out)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t0) / len(cams) * 1e6)
            res.append('%s %.0f us' % ({'0': 'bucket', '1': 'fused'}.get(route, route), sorted(ts)[1]))
        except Exception as exc:                                     # noqa: BLE001
            res.append('%s failed (%s)' % (route, type(exc).__name__))
        oc.pop('_kbePreparedCloud', None)
        torch.cuda.empty_cache()
    print('%.1f points per pixel (%d points): %s per frame' % (n / (size * size), n, ', '.join(res)), flush=True)
