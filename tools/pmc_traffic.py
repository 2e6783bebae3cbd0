#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE), GPU box only.

Runs (1) calibration kernels with a known byte count in THIS code's access pattern (4-byte-per-lane
coalesced streams, buffers larger than the 256 MiB Infinity Cache), then (2) a few frames of the
bench workload.  tools/pmc_report.py turns the two counter CSVs into per-launch HBM bytes.
"""
import os
import sys

os.environ['KBE_FILL_GROUP'] = '1'      # one frame per launch: the per-launch byte counts are per frame

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from ken_burns_effect_amd import _native, common, synthetic  # noqa: E402

K = _native.kernels()
dev = torch.device('cuda:0')
# calibration: k_zkeys_decode reads n*4 B and writes n*4 B with 4 B/lane; k_fill_u32 writes n*4 B
n = 96 * 1024 * 1024          # 384 MiB per buffer
keys = torch.zeros(n, dtype=torch.int32, device=dev)
for _ in range(3):
    K.zkeys_clear(keys)        # k_fill_u32: WRITE n*4
    zee = K.zkeys_decode(keys)     # k_zkeys_decode: READ n*4, WRITE n*4
del keys, zee
torch.cuda.synchronize()

size = int(os.environ.get('SIZE', '1024'))
ofrom, oto = synthetic.default_windows(size, size, False)
settings = {'dblSteps': [i / 15.0 for i in range(16)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': True, 'dolly': False}
oc = bench.build_scene(size, dev, os.environ.get('CLOUD', 'inpaint') == 'inpaint', settings)
cams = common.frame_cameras(settings, oc)
frames = common.render_frames(cams, oc, common.crop_size(settings), keep_on_device=True)      # the route in use at this size
state = common._prepared_cloud(K, oc)
for focal, shift3 in cams:                                                                    # and both routes explicitly, one frame at a time
    K.render_frame(state, shift3, focal, oc['dblBaseline'], fused=True)
    K.render_frame(state, shift3, focal, oc['dblBaseline'], fused=False)
torch.cuda.synchronize()
print('frames', frames.shape, 'points', oc['tensorInpaPoints'].shape[-1])
