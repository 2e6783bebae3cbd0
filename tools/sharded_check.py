#!/usr/bin/env python3
"""process_kenburns_sharded on the HIP path with 2+ ranks: the union of the ranks' frames must equal a single-process
render.  KBE_DIST_BACKEND=gloo (default: the ranks share GPU 0 -- a functional check) or nccl (= RCCL: one GPU per
rank, needs that many GPUs).  torchrun --nproc-per-node 2 tools/sharded_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from ken_burns_effect_amd import common, sharding, synthetic

rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
backend = os.environ.get('KBE_DIST_BACKEND', 'gloo')
index = int(os.environ.get('LOCAL_RANK', rank)) if backend == 'nccl' else 0
torch.cuda.set_device(index)
dev = torch.device('cuda', index)
if backend == 'nccl':
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
else:
    dist.init_process_group('gloo', rank=rank, world_size=world)
H, W = 192, 256
ofrom, oto = synthetic.default_windows(H, W, False)
settings = {'dblSteps': [i / 10.0 for i in range(11)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': False, 'dolly': False}
oc = {}
if rank == 0:
    image, disp = synthetic.make_rgbd(H, W, 4)
    depth = (synthetic.FOCAL * synthetic.BASELINE) / (disp + 1e-7)
    K = common._K()
    oc = {'dblFocal': synthetic.FOCAL, 'dblBaseline': synthetic.BASELINE, 'intWidth': W, 'intHeight': H,
          'objectDepthrange': synthetic.depthrange_of(depth), 'tensorRawImage': image.to(dev), 'tensorRawDisparity': disp.to(dev),
          'tensorRawDepth': depth.to(dev)}
    oc['tensorRawPoints'] = K.depth_to_points(oc['tensorRawDepth'], synthetic.FOCAL).view(1, 3, -1)
    common._reset_inpa(oc)
frames = sharding.process_kenburns_sharded(settings, oc, None, dev, gather=True)
idx, mine = sharding.process_kenburns_sharded(settings, oc, None, dev)           # the default: per-rank delivery to host memory
assert idx == sharding.shard_indices(11, rank, world) and len(mine) == len(idx)
if rank == 0:
    ref = common.process_kenburns(settings, oc, None)
    assert all(np.abs(mine[k].astype(np.int32) - ref[i].astype(np.int32)).max() <= 1 for k, i in enumerate(idx))
    d = np.abs(np.stack(frames).astype(np.int32) - np.stack(ref).astype(np.int32))
    print('sharded vs single process: %d frames, max |diff| %d, differing values %.2e' % (len(frames), d.max(), (d > 0).mean()))
    assert len(frames) == 11 and d.max() <= 1 and (d > 0).mean() < 1e-3
    print('OK (%s, %d ranks)' % (backend, world))
dist.barrier()
dist.destroy_process_group()
