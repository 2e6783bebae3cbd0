#!/usr/bin/env python3
"""Where the point-cloud growth of a 512^2 Pipeline call spends its time (dev aid): the pieces of ONE inpaint pass, each synchronised."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ken_burns_effect_amd import common, kbe, synthetic  # noqa: E402
from ken_burns_effect_amd.pipeline import Pipeline  # noqa: E402

size = int(os.environ.get('SIZE', '512'))
image, _ = synthetic.make_rgbd(size, size, 9)
pipe = Pipeline(model_paths=None, allow_random_weights=True, device='cuda:0', steps=64)
zoom = kbe.windows_for(size, size, dict.fromkeys(('startU', 'startV', 'startW', 'startH', 'endU', 'endV', 'endW', 'endH')), False)
for _ in range(3):
    pipe(image, zoom)
oc, net, K = pipe.objectCommon, pipe.moduleInpaint, common._K()
settings = {'dblSteps': [0.0, 1.0], 'objectFrom': zoom['objectFrom'], 'objectTo': zoom['objectTo'], 'boolInpaint': True, 'dolly': False}


def T(fn, reps=10):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2] * 1e3, r


with torch.no_grad(), common.on_device_of(oc['tensorRawPoints']):
    t_all, _ = T(lambda: common.build_pointcloud(settings, oc, net))
    common._reset_inpa(oc)
    focal, pose = common._camera_at(1.0, settings, oc)
    pose['tensorPoints'] = oc['tensorInpaPoints']
    shift = 1.1 * torch.tensor(common._shift_vector(pose, oc, focal), dtype=torch.float64).to(torch.float32).view(1, 3, 1).cuda()
    img, disp = oc['tensorRawImage'], oc['tensorRawDisparity']
    t_pi, out = T(lambda: net.pointcloud_inpainting(img, disp, shift, oc, focal))
    net._kept_source = None
    t_src, _ = T(lambda: (setattr(net, '_kept_source', None), net.pointcloud_inpainting(img, disp, shift, oc, focal))[1])
    kept = net._kept_source[3]
    t_warp, (render, existing) = T(lambda: K.render_pointcloud(kept[0] + shift, kept[1], size, size, focal, oc['dblBaseline']))
    ex = (existing > 0.0).float()
    t_med, exd = T(lambda: ex * K.spatial_filter(ex, 'median-5'))
    t_fwd, _ = T(lambda: net.forward(tensorData=render * exd, tensorMasks=exd))
    t_proc, _ = T(lambda: (common._reset_inpa(oc), common.process_inpaint(shift, oc, net, focal))[1])
    print('%d^2: build_pointcloud (two passes) %.2f ms; one pass: process_inpaint %.2f = pointcloud_inpainting %.2f (with the source kept; %.2f without) + append; '
          'of it: 68-channel warp %.2f, median-5 + mask %.2f, Inpaint.forward %.2f' % (size, t_all, t_proc, t_pi, t_src, t_warp, t_med, t_fwd))
