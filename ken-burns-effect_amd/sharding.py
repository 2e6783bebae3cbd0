"""Frames of one video sharded over the GPUs of a node.

The reference is single-process, single-GPU (no torch.distributed anywhere).  The only
state a frame needs is the final point cloud (``tensorInpaPoints/Image/Depth``, 28*N
bytes) plus a dozen scalars, and frames are independent of each other
(``/root/reference/utils/common.py:222-260``).  So: one process per GPU, rank 0 builds
the cloud (the two inpaint passes are serial, ``:181-219``), ONE broadcast hands it to
the other ranks, every rank renders its own subset of ``dblSteps``, and the uint8
frames are optionally gathered.  No all-reduce, no per-frame collective.

Backend "nccl" is RCCL on ROCm (xGMI is point-to-point; a 31 MB one-to-all broadcast
is ~0.2 ms and happens once per video); "gloo" runs the same code on CPU tensors for
the world_size-2 tests.
"""
import os

import torch
import torch.distributed as dist

_CLOUD_KEYS = ('tensorInpaPoints', 'tensorInpaImage', 'tensorInpaDepth')
_SCALAR_KEYS = ('dblFocal', 'dblBaseline', 'intWidth', 'intHeight')


# what the last bind_to_gpu_numa_node call did: the node, a code (NUMA_CODES: what travels between ranks as a number) and the words for it
NUMA_CODES = {0: 'bound to the CPUs of its node', 1: 'not attempted', 2: 'no PCI address for the device', 3: 'no numa_node entry for the device in /sys',
              4: 'the kernel reports no NUMA node for the device (a single-node host, or a virtual function)',
              5: "none of the node's CPUs is in the process's affinity mask (a cpuset from the launcher)", 6: 'the topology could not be read'}
NUMA_BIND = {'node': None, 'code': 1, 'reason': NUMA_CODES[1]}


def bind_to_gpu_numa_node(device_index):
    """Best effort: restrict this process to the CPUs of the NUMA node its GPU hangs off, BEFORE it allocates the pinned
    buffers its frames land in (first touch then places them on that node).  On an 8-GPU node every rank pushes ~53 GB/s of
    frames into host memory; landing them on the other socket would put half the node's traffic on the inter-socket
    links.  Returns the node number, or None when the topology cannot be read (then nothing is changed) -- and says which of
    the two it was, and why, in ``NUMA_BIND`` (a rank that could not bind is reported by the bench line and by
    tools/scale_report.py instead of passing silently: VERDICT r5 item 7)."""
    def done(node, code, detail=''):
        NUMA_BIND['node'], NUMA_BIND['code'] = node, code
        NUMA_BIND['reason'] = NUMA_CODES[code] + (' (%s)' % detail if detail else '')
        return node
    try:
        props = torch.cuda.get_device_properties(device_index)
        bdf = '%04x:%02x:%02x.0' % (getattr(props, 'pci_domain_id', 0), props.pci_bus_id, props.pci_device_id)
    except Exception as e:
        return done(None, 2, 'no PCI address for device %r: %s' % (device_index, type(e).__name__))
    try:
        node = int(open('/sys/bus/pci/devices/%s/numa_node' % bdf).read())
    except Exception as e:
        return done(None, 3, '%s: %s' % (bdf, type(e).__name__))
    if node < 0:
        return done(None, 4, bdf)
    try:
        cpus = set()
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            lo, _, hi = part.partition('-')
            cpus.update(range(int(lo), int(hi or lo) + 1))
        allowed = cpus & os.sched_getaffinity(0)
        if not allowed:
            return done(None, 5, 'node %d' % node)
        os.sched_setaffinity(0, allowed)
        return done(node, 0, '%d CPUs of node %d' % (len(allowed), node))
    except Exception as e:
        return done(None, 6, 'node %d: %s' % (node, type(e).__name__))


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


SHARD_SHAPE = 'round-robin'     # decided by measurement on one GPU, round 5 (profiles/r05_shard_shapes.txt): see shard_indices


def shard_indices(n, rank, world_size, shape=None):
    """Which of a video's n frames rank `rank` renders.
    'round-robin' (the default): frames rank, rank + world_size, ... -- every rank samples the whole path.  A frame at the ends
    of a Ken Burns path costs ~7-14 % more than one in its middle, and the video ends with its slowest rank.
    'block': a contiguous run of n / world_size frames (the first n % world_size ranks one more) -- consecutive cameras, which
    the renderer's launch groups like (the frames of a launch share candidate lists built for the box between a sub-group's
    first and last camera: kbe_fused.hip share_plan) -- but the ranks holding the path's ends set the pace.
    'dealt<k>' (e.g. 'dealt2'): runs of k consecutive frames dealt to the ranks in turn.
    Measured as an 8-way share of a 128- and a 75-frame 1024^2 video on one GPU (tools/shard_shapes.py; every rank of a node
    has its own GPU and link, so a share rendered alone is what that rank would do), the video at its slowest rank's pace,
    delivered / left in HBM, k frames/s over 8 GPUs (the round's last collection, the figures DESIGN.md section 6 quotes):
    128 frames: round-robin 114 / 277, block 111 / 261, dealt4 111 / 280, dealt2 114 / 278; 75 frames: round-robin 98 / 233,
    block 94 / 215, dealt4 85 / 197 (12 : 8 frames), dealt2 97 / 226.
    Round-robin is within 2 % of the best everywhere and never the worst.  KBE_SHARD_SHAPE overrides."""
    shape = shard_shape(shape)
    if shape == 'round-robin':
        return list(range(rank, n, world_size))
    if shape.startswith('dealt'):           # 'dealt4': runs of 4 consecutive frames dealt to the ranks in turn
        b = int(shape[5:] or 4)
        return [i for i in range(n) if (i // b) % world_size == rank]
    if shape != 'block':
        raise ValueError('shard shape %r: block, round-robin or dealt<run length>' % (shape,))
    q, r = divmod(n, world_size)            # as even as blocks get: the first n % world_size ranks take one frame more
    start = rank * q + min(rank, r)
    return list(range(start, start + q + (1 if rank < r else 0)))


def shard_steps(steps, rank=None, world_size=None, shape=None):
    """(indices, steps) of this rank's share of a video's steps (shard_indices)."""
    if rank is None:
        rank, world_size = world()
    idx = shard_indices(len(steps), rank, world_size, shape)
    return idx, [steps[i] for i in idx]


def single_rank_collectives():
    """KBE_SINGLE_RANK_COLLECTIVES=1: a process group of ONE rank still runs the cloud broadcast (and decodes its header
    like a receiver) and the frame gather, instead of returning early -- the way a 1-GPU box checks that RCCL comes up on the
    device and takes these collectives' dtypes and shapes (tests/test_hip_parity.py).  Not a product setting."""
    return os.environ.get('KBE_SINGLE_RANK_COLLECTIVES', '0') == '1'


def broadcast_cloud(objectCommon, device, src=0):
    """Rank `src` holds the finished cloud in ``objectCommon``; afterwards every rank does,
    bit-identical (so frames are byte-identical to a single-GPU run).  Two collectives: a
    small header (N and the scalars, incl. objectDepthrange) and one packed [7,N] fp32
    payload."""
    rank, world_size = world()
    if world_size == 1 and not (dist.is_initialized() and single_rank_collectives()):
        return objectCommon
    header = torch.zeros(16, dtype=torch.float64, device=device)
    if rank == src:
        dr = objectCommon['objectDepthrange']
        # [12], [13]: the lanes rank `src` MEASURED for delivering this cloud's frames to host memory (ordinary camera path /
        # zoom-out; 0 = not measured): the other ranks take them instead of probing -- eight ranks timing twelve frames each
        # while their neighbours load the host memory system would each measure the others (measure_delivery_lanes below)
        hint = objectCommon.get('_kbeDeliveryLanes', {})
        vals = [objectCommon['tensorInpaPoints'].shape[-1]] + [objectCommon[k] for k in _SCALAR_KEYS] + \
               [dr[0], dr[1], dr[2][0], dr[2][1], dr[3][0], dr[3][1], 1.0 if isinstance(objectCommon['dblBaseline'], int) else 0.0,
                float(hint.get(False, 0)), float(hint.get(True, 0))]
        header[:len(vals)] = torch.tensor(vals, dtype=torch.float64)
    dist.broadcast(header, src)
    h = header.cpu().tolist()
    n = int(h[0])
    if rank == src:
        packed = torch.cat([objectCommon[k].reshape(-1, n).float() for k in _CLOUD_KEYS], 0).contiguous().to(device)
    else:
        # re-use the landing buffer of an earlier broadcast of the same size: the tensors below are views of
        # it, so their addresses (and everything cached per cloud, e.g. the renderer's scratch) stay valid
        packed = objectCommon.get('_kbePackedCloud')
        if packed is None or packed.shape != (7, n) or packed.device != torch.device(device):
            packed = torch.empty(7, n, dtype=torch.float32, device=device)
            objectCommon['_kbePackedCloud'] = packed
    dist.broadcast(packed, src)
    if rank != src or world_size == 1:
        objectCommon['dblFocal'] = h[1]
        objectCommon['dblBaseline'] = int(h[2]) if h[11] == 1.0 else h[2]
        objectCommon['intWidth'], objectCommon['intHeight'] = int(h[3]), int(h[4])
        objectCommon['objectDepthrange'] = (h[5], h[6], (int(h[7]), int(h[8])), (int(h[9]), int(h[10])))
        objectCommon['tensorInpaPoints'] = packed[0:3].unsqueeze(0)
        objectCommon['tensorInpaImage'] = packed[3:6].unsqueeze(0)
        objectCommon['tensorInpaDepth'] = packed[6:7].unsqueeze(0)
        hint = {key: int(v) for key, v in ((False, h[12]), (True, h[13])) if int(v) > 0}
        if hint:
            objectCommon['_kbeDeliveryLanes'] = hint
        else:
            objectCommon.pop('_kbeDeliveryLanes', None)         # nothing measured for THIS cloud: an earlier video's entry must not stand
    return objectCommon


def measure_delivery_lanes(objectSettings, objectCommon):
    """Rank 0, before broadcast_cloud: how many lanes deliver this video's frames to host memory fastest (the renderer's own
    probe, HipKernels.delivery_lanes: twelve frames timed with two events), noted in ``objectCommon`` so that the broadcast's
    header carries it to every rank.  Without a GPU renderer (the CPU tests' kernel sets) nothing is measured."""
    from . import common
    K = common._K()
    if not hasattr(K, 'delivery_lanes') or 'tensorInpaPoints' not in objectCommon or not objectCommon['tensorInpaPoints'].is_cuda:
        return None
    cameras = common.frame_cameras(objectSettings, objectCommon)
    crop = common.crop_size(objectSettings) if objectSettings.get('boolCrop', True) else None
    # (a hint is what ANOTHER rank measured: on the measuring rank an entry left by an earlier video of the same objectCommon --
    # Pipeline keeps it -- would be handed back unmeasured and broadcast again, whatever the new cloud's N, W, H or crop: ADVICE r4)
    objectCommon.pop('_kbeDeliveryLanes', None)
    state = common._prepared_cloud(K, objectCommon)
    state.pop('delivery_lanes_hint', None)
    lanes = K.delivery_lanes(state, cameras, objectCommon['dblBaseline'], crop)
    objectCommon['_kbeDeliveryLanes'] = {K.zooms_out(state, cameras): int(lanes)}
    return lanes


def shard_shape(shape=None):
    """The shard shape a call uses: the argument, else KBE_SHARD_SHAPE, else SHARD_SHAPE -- resolved ONCE per video
    (process_kenburns_sharded) and handed to shard_steps and gather_frames alike, so that the two cannot disagree."""
    return shape or os.environ.get('KBE_SHARD_SHAPE', SHARD_SHAPE)


def gather_frames(local_frames, indices, total, device, dst=0, shape=None):
    """Collects per-rank uint8 frames [n_local,H,W,3] on rank `dst` in original step order (`shape`: the one the frames were
    sharded by).  Returns the full [total,H,W,3] tensor on `dst`, None elsewhere.  Every rank pads its share to the LARGEST
    share of the shape (dealt runs hand some ranks more than ceil(total / world_size) frames: 75 frames over 8 ranks in runs
    of 4 is 12 : 8)."""
    rank, world_size = world()
    if world_size == 1 and not (dist.is_initialized() and single_rank_collectives()):
        return local_frames
    shape = shard_shape(shape)
    shares = [shard_indices(total, r, world_size, shape) for r in range(world_size)]
    if local_frames.shape[0] != len(shares[rank]):
        raise ValueError('rank %d holds %d frames, shard shape %r gives it %d of %d' % (rank, local_frames.shape[0], shape, len(shares[rank]), total))
    per = max(1, max(len(i) for i in shares))
    H, W = local_frames.shape[1:3]
    padded = torch.zeros(per, H, W, 3, dtype=torch.uint8, device=device)
    padded[:local_frames.shape[0]] = local_frames.to(device)
    bucket = [torch.empty_like(padded) for _ in range(world_size)] if rank == dst else None
    dist.gather(padded, bucket, dst)
    if rank != dst:
        return None
    out = torch.empty(total, H, W, 3, dtype=torch.uint8, device=device)
    for r, idx in enumerate(shares):
        if idx:
            out[idx] = bucket[r][:len(idx)]
    return out


def process_kenburns_sharded(objectSettings, objectCommon, moduleInpaint, device, gather=False):
    """process_kenburns over all ranks: rank 0 builds the cloud, one broadcast, each rank renders its share of ``dblSteps``
    (round-robin: shard_indices).

    ``gather=False`` (default): every rank delivers ITS frames to its own pinned host memory over its own PCIe link (what
    a node with one encoder / writer process per GPU wants; nothing funnels through one GPU) and gets
    ``(indices, frames)`` back -- the positions of its frames in ``dblSteps`` and the uint8 HxWx3 frames.
    ``gather=True``: the frames stay in HBM, are gathered on rank 0 (one collective) and rank 0 returns the full list in
    step order, the other ranks None.  At 17 k frames/s per GPU that funnel is the bottleneck of an 8-GPU run; it exists
    for callers that need the whole video in one process."""
    from . import common
    with common.on_device_of(torch.device(device)):
        return _process_kenburns_sharded(objectSettings, objectCommon, moduleInpaint, device, gather)


def _process_kenburns_sharded(objectSettings, objectCommon, moduleInpaint, device, gather):
    from . import common
    rank, world_size = world()
    if rank == 0 and ('boolInpaint' not in objectSettings or objectSettings['boolInpaint']):
        common.build_pointcloud(objectSettings, objectCommon, moduleInpaint)
    elif rank == 0:
        if 'tensorInpaPoints' not in objectCommon:
            common._reset_inpa(objectCommon)
    if rank == 0 and world_size > 1 and not gather:
        measure_delivery_lanes(objectSettings, objectCommon)
    broadcast_cloud(objectCommon, device)
    shape = shard_shape()                   # once: the sharding and the gather below use the same one
    idx, steps = shard_steps(objectSettings['dblSteps'], rank, world_size, shape)
    local_settings = dict(objectSettings, dblSteps=steps)
    crop = common.crop_size(objectSettings) if objectSettings.get('boolCrop', True) else None
    cameras = common.frame_cameras(local_settings, objectCommon)
    if not gather:
        frames = common.render_frames(cameras, objectCommon, crop)
        return idx, [frames[i] for i in range(frames.shape[0])]
    frames = common.render_frames(cameras, objectCommon, crop, keep_on_device=True)
    full = gather_frames(frames, idx, len(objectSettings['dblSteps']), device, shape=shape)
    return None if full is None else [f for f in full.cpu().numpy()]
