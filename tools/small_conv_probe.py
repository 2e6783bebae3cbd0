#!/usr/bin/env python3
"""The Disparity GridNet's 512-channel rows at 32^2 / 16^2 / 8^2 pixels: F.conv2d (MIOpen, tuned by one find step) against the same
convolution as unfold + matmul (dev aid)."""
import time
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device('cuda:0')
w = torch.randn(512, 512, 3, 3, device=dev) * 0.02
wm = w.view(512, -1).contiguous()
b = torch.randn(512, device=dev)


def T(fn, n=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


with torch.no_grad():
    for s in (32, 16, 8, 4):
        x = torch.randn(1, 512, s, s, device=dev)
        conv = lambda: F.conv2d(x, w, None, 1, 1)
        gemm = lambda: torch.matmul(wm, F.unfold(x, 3, padding=1)[0]).view(1, 512, s, s)
        err = float((conv() - gemm()).abs().max() / conv().abs().max())
        print('%2d^2: conv2d %.1f us, unfold + matmul %.1f us (unfold alone %.1f), relative difference %.2g' % (s, T(conv), T(gemm), T(lambda: F.unfold(x, 3, padding=1)), err))
