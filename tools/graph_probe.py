#!/usr/bin/env python3
"""Is the GridNet forward bound by its launches?  Inpaint.forward eager against a captured HIP graph replayed (dev aid)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ken_burns_effect_amd import synthetic
from ken_burns_effect_amd.pointcloud_inpainting import Inpaint
dev = torch.device('cuda:0')
for size in (256, 512, 1024):
    net = synthetic.seeded_fill_(Inpaint(), 3).to(dev).eval()
    data = torch.randn(1, 68, size, size, device=dev); mask = torch.ones(1, 1, size, size, device=dev)
    with torch.no_grad():
        net.normalize_images_disp(torch.rand(1, 3, size, size, device=dev), torch.rand(1, 1, size, size, device=dev), not_normed=True)
        def run():
            return net.forward(tensorData=data, tensorMasks=mask)
        for _ in range(3): run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): out = run()
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 10 * 1e3
        # host time of the enqueue alone
        t0 = time.perf_counter()
        for _ in range(10): out = run()
        enq = (time.perf_counter() - t0) / 10 * 1e3
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): run()
        torch.cuda.current_stream().wait_stream(s)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                gout = run()
            g.replay(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10): g.replay()
            torch.cuda.synchronize(); graph = (time.perf_counter() - t0) / 10 * 1e3
            same = float((gout['tensorImage'] - out['tensorImage']).abs().max())
            print('size %4d: eager %.2f ms (host enqueue %.2f ms), graph replay %.2f ms, max |diff| %.3g' % (size, eager, enq, graph, same))
        except Exception as e:
            print('size %4d: eager %.2f ms (host enqueue %.2f ms), capture failed: %s' % (size, eager, enq, str(e)[:200]))
