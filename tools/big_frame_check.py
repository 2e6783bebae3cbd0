#!/usr/bin/env python3
"""One frame at a size beyond 4096^2 (64-bit bucket offsets in k_project), tile renderer against the stage-by-stage
global-atomic HIP path (GPU box only; dev aid).   SIZE=5120 python tools/big_frame_check.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

size = int(os.environ.get('SIZE', '5120'))
K = _native.kernels()
dev = torch.device('cuda:0')
image, disp = synthetic.make_rgbd(size, size, seed=0)
depth = ((synthetic.FOCAL * synthetic.BASELINE) / (disp + 1e-7)).to(dev)
pts = K.depth_to_points(depth, synthetic.FOCAL).view(1, 3, -1)
img = image.to(dev).reshape(1, 3, -1)
dep = depth.reshape(1, 1, -1)
shift3 = (size * 0.004, -size * 0.003, -size * 0.02)
state = K.prepare_cloud(pts, img, dep, size, size, raster=(size, size * size))
rf = torch.empty(4, size, size, device=dev)
ex = torch.empty(size * size, device=dev)
frame = K.render_frame(state, shift3, synthetic.FOCAL, synthetic.BASELINE, render_f32=rf, existing_f32=ex)
p2 = K.shift_points(pts, shift3)
render, existing = K.render_pointcloud(p2, torch.cat([img, dep], 1), size, size, synthetic.FOCAL, synthetic.BASELINE, tiled=False)
filled = K.fill_disocclusion(render, render[:, 3:4] * (existing > 0.0).float())
torch.cuda.synchronize()
same_mask = bool(torch.equal(ex.view(size, size) > 0, existing[0, 0] > 0))
err = float((rf - filled[0]).abs().max()) / max(1.0, float(filled.abs().max()))
du8 = int((frame.int() - K.frame_u8(filled).int()).abs().max())
print('size %d: validity masks equal %s, max relative float difference %.2e, max uint8 difference %d, holes %d' % (
    size, same_mask, err, du8, int((existing[0, 0] <= 0).sum())))
assert same_mask and err < 1e-4 and du8 <= 1
