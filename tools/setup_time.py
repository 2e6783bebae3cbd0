import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
import bench
from ken_burns_effect_amd import _native, common, synthetic
from ken_burns_effect_amd.pointcloud_inpainting import Inpaint
dev = torch.device('cuda:0')
size = 1024
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [0.0, 1.0], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, dev, False, settings)
net = synthetic.seeded_fill_(Inpaint(), 3).to(dev).eval()
K = _native.kernels()
def T(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print('build_pointcloud (2 inpaint passes): %.1f ms' % T(lambda: (common._reset_inpa(oc), common.build_pointcloud(settings, oc, net)), 3))
    pts = oc['tensorRawPoints']; feat = torch.randn(1, 68, size * size, device=dev)
    print('render_pointcloud 68 ch: %.2f ms' % T(lambda: K.render_pointcloud(pts, feat, size, size, 512.0, 120)))
    img = oc['tensorRawImage']; disp = oc['tensorRawDisparity']
    data = torch.randn(1, 68, size, size, device=dev); mask = torch.ones(1, 1, size, size, device=dev)
    print('Inpaint.forward: %.1f ms' % T(lambda: net.forward(tensorData=data, tensorMasks=mask)))
    print('moduleContext: %.1f ms' % T(lambda: net.moduleContext(torch.cat([img, disp], 1))))
    print('median-5: %.2f ms' % T(lambda: K.spatial_filter(mask, 'median-5')))
