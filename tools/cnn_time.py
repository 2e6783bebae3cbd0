import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ken_burns_effect_amd import synthetic
from ken_burns_effect_amd.pointcloud_inpainting import Inpaint
if os.environ.get('NET') == 'partial':
    from ken_burns_effect_amd.partial_inpainting import Inpaint
dev = torch.device('cuda:0')
size = int(os.environ.get('SIZE', '1024'))
if os.environ.get('BENCHMARK') == '1':
    torch.backends.cudnn.benchmark = True
net = synthetic.seeded_fill_(Inpaint(), 3).to(dev).eval()
net.compute_dtype = {'bf16': torch.bfloat16, 'fp16': torch.float16}.get(os.environ.get('DTYPE', ''))
data = torch.randn(1, 68, size, size, device=dev); mask = (torch.rand(1, 1, size, size, device=dev) > 0.2).float()
if os.environ.get('CL') == '1':
    net = net.to(memory_format=torch.channels_last); data = data.contiguous(memory_format=torch.channels_last)
def T(fn, n=5):
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); first = (time.perf_counter() - t0) * 1e3
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return first, (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    net.normalize_images_disp(torch.rand(1, 3, size, size, device=dev), torch.rand(1, 1, size, size, device=dev), not_normed=True)
    first, steady = T(lambda: net.forward(tensorData=data, tensorMasks=mask))
    ref = os.environ.get('DTYPE')
    print('FIND=%s BENCHMARK=%s CL=%s DTYPE=%s: first call %.0f ms, steady %.1f ms' % (os.environ.get('MIOPEN_FIND_MODE'), os.environ.get('BENCHMARK'), os.environ.get('CL'), ref, first, steady))
    out = net.forward(tensorData=data, tensorMasks=mask)
    net.compute_dtype = None
    base = net.forward(tensorData=data, tensorMasks=mask)
    print('   max |image - fp32 image| = %.4g, max |disparity - fp32| = %.4g (relative to max %.4g)' % (float((out['tensorImage'] - base['tensorImage']).abs().max()), float((out['tensorDisparity'] - base['tensorDisparity']).abs().max()), float(base['tensorDisparity'].abs().max())))
