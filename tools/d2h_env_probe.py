#!/usr/bin/env python3
"""Which engine moves a device-to-pinned-host hipMemcpyAsync (dev aid): times 24 MB copies alone and next to a compute kernel;
run under rocprofv3 --kernel-trace --memory-copy-trace to see whether __amd_rocclr_copyBuffer (a blit kernel) or an SDMA entry appears."""
import os
import time

import torch

dev = torch.device('cuda:0')
n = 8 * 1024 * 1024 * 3
src = torch.randint(0, 255, (n,), dtype=torch.uint8, device=dev)
dst = torch.zeros(n, dtype=torch.uint8, pin_memory=True)
a = torch.randn(4096, 4096, device=dev)
s2 = torch.cuda.Stream()
for _ in range(3):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dst.copy_(src, non_blocking=True)
torch.cuda.synchronize()
alone = (time.perf_counter() - t0) / 20
b = a @ a
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    b = a @ a
torch.cuda.synchronize()
mm = (time.perf_counter() - t0) / 20
t0 = time.perf_counter()
for _ in range(20):
    with torch.cuda.stream(s2):
        dst.copy_(src, non_blocking=True)
    b = a @ a
torch.cuda.synchronize()
both = (time.perf_counter() - t0) / 20
print('%s: copy alone %.0f us (%.1f GB/s), matmul alone %.0f us, both concurrently %.0f us per pair (ok %s)' % (
    ' '.join('%s=%s' % (k, os.environ[k]) for k in ('GPU_BLIT_ENGINE_TYPE', 'GPU_FORCE_BLIT_COPY_SIZE', 'HSA_ENABLE_SDMA', 'DEBUG_CLR_LIMIT_BLIT_WG', 'GPU_PINNED_MIN_XFER_SIZE', 'ROC_P2P_SDMA_SIZE') if k in os.environ) or 'default',
    alone * 1e6, n / alone / 1e9, mm * 1e6, both * 1e6, bool((dst.to(dev) == src).all())))
