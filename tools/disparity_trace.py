#!/usr/bin/env python3
"""The Disparity network's forward (a 512^2 image: 256^2 input) for rocprofv3 --kernel-trace --stats (dev aid)."""
import os, sys
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import torch
from ken_burns_effect_amd import synthetic
from ken_burns_effect_amd.disparity_estimation import Disparity, Semantics
from ken_burns_effect_amd.utils import resize_image
dev = torch.device('cuda:0')
size = int(os.environ.get('SIZE', '512'))
sem = synthetic.seeded_fill_(Semantics(), 1).to(dev).eval()
dis = synthetic.seeded_fill_(Disparity(), 2).to(dev).eval()
image = torch.rand(1, 3, size, size, device=dev)
with torch.no_grad():
    resized = resize_image(image, max_size=size // 2)
    feat = sem(resized)
    for _ in range(12):
        out = dis(resized, feat)
torch.cuda.synchronize()
print(out.shape)
