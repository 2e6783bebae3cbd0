"""Frame sharding over ranks (world_size 2, gloo, CPU): one broadcast of the cloud, round-robin frames,
gather -- the union must equal the single-process frame list byte for byte.  Kernel set = the oracle
(injected explicitly in every rank; the product path itself is HIP-only)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _scene():
    sys.path.insert(0, ROOT)
    from ken_burns_effect_amd import synthetic
    from oracle import kbe_oracle
    H, W = 40, 56
    image, disp = synthetic.make_rgbd(H, W, 61)
    depth = (512.0 * 120) / (disp + 1e-7)
    oc = {'dblFocal': 512.0, 'dblBaseline': 120, 'intWidth': W, 'intHeight': H, 'objectDepthrange': synthetic.depthrange_of(depth),
          'tensorRawImage': image, 'tensorRawDisparity': disp, 'tensorRawDepth': depth,
          'tensorRawPoints': kbe_oracle.depth_to_points(depth, 512.0).view(1, 3, -1)}
    ofrom, oto = synthetic.default_windows(H, W, False)
    settings = {'dblSteps': [i / 6.0 for i in range(7)], 'objectFrom': ofrom, 'objectTo': oto, 'boolInpaint': False, 'dolly': False}
    return settings, oc


def _worker(rank, world_size, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    try:
        from ken_burns_effect_amd import common, sharding
        from oracle import kbe_oracle
        common._kernel_set = kbe_oracle.OracleKernels('jacobi')
        settings, oc = _scene()
        if rank != 0:
            oc = {}                    # only rank 0 owns the scene; the others learn it from the broadcast
        else:
            common._reset_inpa(oc)
        frames = sharding.process_kenburns_sharded(settings, oc, None, torch.device('cpu'), gather=True)
        idx, _ = sharding.shard_steps(settings['dblSteps'], rank, world_size)
        assert oc['tensorInpaPoints'].shape == (1, 3, 40 * 56) and oc['intWidth'] == 56 and isinstance(oc['dblBaseline'], int)
        # the default: every rank keeps (delivers) its own frames; rank 0's must be its rows of the gathered video
        mine_idx, mine = sharding.process_kenburns_sharded(dict(settings, boolInpaint=False), oc, None, torch.device('cpu'))
        assert mine_idx == idx and len(mine) == len(idx) and mine[0].shape == (40, 56, 3) and mine[0].dtype == np.uint8
        if rank == 0:
            np.save(out_path, np.stack(frames))
            assert all(np.array_equal(frames[i], f) for i, f in zip(idx, mine))
        else:
            assert frames is None and len(idx) == 3
    finally:
        dist.destroy_process_group()


def test_two_ranks_render_the_same_video_as_one(tmp_path):
    sys.path.insert(0, ROOT)
    from ken_burns_effect_amd import common
    from oracle import kbe_oracle
    out = str(tmp_path / 'frames.npy')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    sharded = np.load(out)
    common._kernel_set = kbe_oracle.OracleKernels('jacobi')
    try:
        settings, oc = _scene()
        common._reset_inpa(oc)
        single = np.stack(common.process_kenburns(settings, oc, None))
    finally:
        common._kernel_set = None
    assert sharded.shape == single.shape == (7, 40, 56, 3)
    assert np.array_equal(sharded, single)


def _worker_many(rank, world_size, port, out_dir, n_frames):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1')
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    try:
        from ken_burns_effect_amd import common, sharding
        from oracle import kbe_oracle
        common._kernel_set = kbe_oracle.OracleKernels('jacobi')
        settings, oc = _scene()
        settings['dblSteps'] = [i / (n_frames - 1.0) for i in range(n_frames)]
        if rank != 0:
            oc = {}
        else:
            common._reset_inpa(oc)
            oc['_kbeDeliveryLanes'] = {False: 3}        # what rank 0 measured travels with the cloud's header
        idx, mine = sharding.process_kenburns_sharded(settings, oc, None, torch.device('cpu'))       # every rank keeps its frames
        assert oc['_kbeDeliveryLanes'] == {False: 3}
        assert idx == sharding.shard_indices(n_frames, rank, world_size) == list(range(rank, n_frames, world_size)) and len(mine) == len(idx)
        np.savez(os.path.join(out_dir, 'rank%d.npz' % rank), idx=np.array(idx), frames=np.stack(mine))
        full = sharding.process_kenburns_sharded(settings, oc, None, torch.device('cpu'), gather=True)
        if rank == 0:
            np.save(os.path.join(out_dir, 'gathered.npy'), np.stack(full))
        else:
            assert full is None
    finally:
        dist.destroy_process_group()


def test_eight_ranks_render_a_75_frame_video_as_one_process_does(tmp_path):
    """The product's video length on a node's worth of ranks (gloo, CPU, the oracle as the kernel set): 75 frames over 8 ranks --
    an uneven 10 / 9 split -- every rank keeping its own frames (the default) and gathered on rank 0: both byte-identical to one
    process rendering the same steps."""
    sys.path.insert(0, ROOT)
    from ken_burns_effect_amd import common
    from oracle import kbe_oracle
    n_frames, world_size = 75, 8
    mp.spawn(_worker_many, args=(world_size, _free_port(), str(tmp_path), n_frames), nprocs=world_size, join=True)
    common._kernel_set = kbe_oracle.OracleKernels('jacobi')
    try:
        settings, oc = _scene()
        settings['dblSteps'] = [i / (n_frames - 1.0) for i in range(n_frames)]
        common._reset_inpa(oc)
        single = np.stack(common.process_kenburns(settings, oc, None))
    finally:
        common._kernel_set = None
    kept = np.zeros_like(single)
    counts = []
    for r in range(world_size):
        z = np.load(str(tmp_path / ('rank%d.npz' % r)))
        kept[z['idx']] = z['frames']
        counts.append(len(z['idx']))
    assert counts == [10, 10, 10, 9, 9, 9, 9, 9]
    assert np.array_equal(kept, single), 'frames the ranks kept'
    assert np.array_equal(np.load(str(tmp_path / 'gathered.npy')), single), 'frames gathered on rank 0'


def _worker_shape(rank, world_size, port, out_dir, n_frames, shape):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), OMP_NUM_THREADS='1', KBE_SHARD_SHAPE=shape)
    sys.path.insert(0, ROOT)
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    try:
        from ken_burns_effect_amd import common, sharding
        from oracle import kbe_oracle
        common._kernel_set = kbe_oracle.OracleKernels('jacobi')
        settings, oc = _scene()
        settings['dblSteps'] = [i / (n_frames - 1.0) for i in range(n_frames)]
        if rank != 0:
            oc = {}
        else:
            common._reset_inpa(oc)
        full = sharding.process_kenburns_sharded(settings, oc, None, torch.device('cpu'), gather=True)
        if rank == 0:
            np.save(os.path.join(out_dir, 'gathered_%s.npy' % shape), np.stack(full))
        else:
            assert full is None
        # a share that is not what the shape gives this rank is refused, not padded into the wrong rows
        if rank == 0:
            with pytest.raises(ValueError):
                sharding.gather_frames(torch.zeros(1, 4, 4, 3, dtype=torch.uint8), [0], n_frames, torch.device('cpu'), shape=shape)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('shape,largest', [('block', 5), ('dealt4', 7), ('dealt2', 6)])
def test_gathered_video_under_every_shard_shape(tmp_path, shape, largest):
    """the gather padded every rank's share to ceil(total / world_size) frames, which a dealt shape exceeds (19 frames
    over 4 ranks in runs of 4: 7 : 4 : 4 : 4 against ceil = 5) -- the padded copy then failed on the shapes.  The pad is now the
    LARGEST share of the shape, resolved once per video; the gathered video equals one process's under every shape."""
    sys.path.insert(0, ROOT)
    from ken_burns_effect_amd import common, sharding
    from oracle import kbe_oracle
    n_frames, world_size = 19, 4
    assert max(len(sharding.shard_indices(n_frames, r, world_size, shape)) for r in range(world_size)) == largest
    assert sorted(i for r in range(world_size) for i in sharding.shard_indices(n_frames, r, world_size, shape)) == list(range(n_frames))
    mp.spawn(_worker_shape, args=(world_size, _free_port(), str(tmp_path), n_frames, shape), nprocs=world_size, join=True)
    common._kernel_set = kbe_oracle.OracleKernels('jacobi')
    try:
        settings, oc = _scene()
        settings['dblSteps'] = [i / (n_frames - 1.0) for i in range(n_frames)]
        common._reset_inpa(oc)
        single = np.stack(common.process_kenburns(settings, oc, None))
    finally:
        common._kernel_set = None
    assert np.array_equal(np.load(str(tmp_path / ('gathered_%s.npy' % shape))), single)


def _worker_one(rank, world_size, port, out_path):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), KBE_SINGLE_RANK_COLLECTIVES='1')
    sys.path.insert(0, ROOT)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    try:
        from ken_burns_effect_amd import common, sharding
        from oracle import kbe_oracle
        common._kernel_set = kbe_oracle.OracleKernels('jacobi')
        settings, oc = _scene()
        common._reset_inpa(oc)
        before = {k: oc[k].clone() for k in ('tensorInpaPoints', 'tensorInpaImage', 'tensorInpaDepth')}
        scalars = (oc['dblFocal'], oc['dblBaseline'], oc['intWidth'], oc['intHeight'], oc['objectDepthrange'])
        frames = sharding.process_kenburns_sharded(settings, oc, None, torch.device('cpu'), gather=True)
        # the receiver's decode ran on this rank: the cloud and its scalars came back as they went in
        assert all(torch.equal(before[k], oc[k]) for k in before) and '_kbePackedCloud' not in oc
        assert (oc['dblFocal'], oc['dblBaseline'], oc['intWidth'], oc['intHeight']) == scalars[:4] and type(oc['dblBaseline']) is type(scalars[1])
        assert tuple(oc['objectDepthrange'][:2]) == tuple(scalars[4][:2]) and tuple(map(tuple, oc['objectDepthrange'][2:])) == tuple(map(tuple, scalars[4][2:]))
        np.save(out_path, np.stack(frames))
    finally:
        dist.destroy_process_group()


def test_a_group_of_one_rank_can_be_made_to_run_the_collectives(tmp_path):
    """KBE_SINGLE_RANK_COLLECTIVES=1 (the mode the GPU suite uses to put the broadcast and the gather through RCCL on a 1-GPU
    box): header and payload broadcast, the receiver's decode, the frame gather -- same frames as without a process group."""
    sys.path.insert(0, ROOT)
    from ken_burns_effect_amd import common
    from oracle import kbe_oracle
    out = str(tmp_path / 'frames.npy')
    mp.spawn(_worker_one, args=(1, _free_port(), out), nprocs=1, join=True)
    common._kernel_set = kbe_oracle.OracleKernels('jacobi')
    try:
        settings, oc = _scene()
        common._reset_inpa(oc)
        single = np.stack(common.process_kenburns(settings, oc, None))
    finally:
        common._kernel_set = None
    assert np.array_equal(np.load(out), single)


def test_shard_steps_partition():
    from ken_burns_effect_amd import sharding
    for shape in (None, 'block', 'round-robin', 'dealt2'):
        for n, ws in ((10, 4), (75, 8), (128, 8), (5, 8), (0, 3), (7, 1)):
            steps = list(range(100, 100 + n))
            seen, sizes = [], []
            for r in range(ws):
                idx, mine = sharding.shard_steps(steps, r, ws, shape)
                assert mine == [steps[i] for i in idx] and idx == sharding.shard_indices(n, r, ws, shape)
                seen += idx
                sizes.append(len(idx))
                if shape == 'block':
                    assert idx == list(range(idx[0], idx[0] + len(idx))) if idx else True, 'a block is a contiguous run of frames'
            assert sorted(seen) == list(range(n)) and (max(sizes) - min(sizes) <= 1 or shape == 'dealt2')
    assert sharding.shard_steps(list(range(10)), 0, 1)[0] == list(range(10))
    assert sharding.SHARD_SHAPE == 'round-robin' and sharding.shard_indices(75, 7, 8, 'block') == list(range(66, 75))
    assert sharding.shard_indices(10, 1, 4) == [1, 5, 9] and sharding.shard_indices(10, 1, 4, 'dealt2') == [2, 3]
