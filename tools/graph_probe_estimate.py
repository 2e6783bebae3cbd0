#!/usr/bin/env python3
"""Are the estimation networks (Semantics + Disparity + Refine on a 512^2 image: layers down to 16^2) bound by their launches?
Each network eager against a captured HIP graph replayed (dev aid)."""
import os, sys, time
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ken_burns_effect_amd import synthetic
from ken_burns_effect_amd.disparity_estimation import Disparity, Semantics
from ken_burns_effect_amd.disparity_refinement import Refine
from ken_burns_effect_amd.utils import resize_image
dev = torch.device('cuda:0')
sem = synthetic.seeded_fill_(Semantics(), 1).to(dev).eval()
dis = synthetic.seeded_fill_(Disparity(), 2).to(dev).eval()
ref = synthetic.seeded_fill_(Refine(), 3).to(dev).eval()


def probe(name, run):
    with torch.no_grad():
        for _ in range(3): run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): out = run()
        torch.cuda.synchronize(); eager = (time.perf_counter() - t0) / 10 * 1e3
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
            print('%-28s eager %.2f ms (host enqueue %.2f ms), graph replay %.2f ms, max |diff| %.3g' % (name, eager, enq, graph, float((gout - out).abs().max())), flush=True)
        except Exception as e:
            print('%-28s eager %.2f ms (host enqueue %.2f ms), capture failed: %s' % (name, eager, enq, str(e).split(chr(10))[0][:200]), flush=True)
            torch.cuda.synchronize()


for size in (512,):
    image = torch.rand(1, 3, size, size, device=dev)
    with torch.no_grad():
        resized = resize_image(image, max_size=size // 2)
        feat = sem(resized)
        coarse = dis(resized, feat)
    probe('resize %d' % size, lambda: resize_image(image, max_size=size // 2))
    probe('Refine %d' % size, lambda: ref(image, coarse))
    probe('Disparity %d' % size, lambda: dis(resized, feat))
    probe('Semantics %d' % size, lambda: sem(resized))
