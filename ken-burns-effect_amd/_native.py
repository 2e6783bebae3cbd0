"""ctypes binding of ``csrc/libkbe_hip.so`` (C ABI: ``include/kbe.h``).

This is the only compute back end of the package.  There is deliberately no CPU or
PyTorch fallback: :func:`load` raises if the library is missing, and every wrapper
refuses tensors that are not contiguous fp32 CUDA(=HIP) tensors.

The wrappers take torch tensors only to obtain ``data_ptr()`` and the current HIP
stream -- the same contract the reference had with CuPy
(``/root/reference/utils/common.py:516-521``), minus the per-shape JIT.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('KBE_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libkbe_hip.so')      # (KBE_LIB_PATH: a variant build of the same ABI -- measurements)

# every symbol include/kbe.h declares (tests check the library exports exactly these)
SYMBOLS = (
    'kbe_abi_version', 'kbe_last_error', 'kbe_device_info', 'kbe_selftest_err', 'kbe_selftest_division', 'kbe_zkeys_clear', 'kbe_zsplat', 'kbe_zkeys_decode',
    'kbe_degrid', 'kbe_degrid_serial', 'kbe_accumulate', 'kbe_normalize', 'kbe_render_pointcloud', 'kbe_fill_disocclusion',
    'kbe_frame_scratch_bytes', 'kbe_frame_scratch_init', 'kbe_render_frame', 'kbe_render_frame_stages', 'kbe_render_frame_group', 'kbe_cloud_pack_bytes', 'kbe_cloud_pack', 'kbe_render_frame_fused', 'kbe_render_frame_group_fused', 'kbe_render_frame_group_ahead_ok', 'kbe_render_frame_group_ahead', 'kbe_video_scratch_stride', 'kbe_video_stage_bytes', 'kbe_render_video', 'kbe_render_pointcloud_tiled', 'kbe_generate_mask', 'kbe_frame_u8', 'kbe_crop_resize_u8', 'kbe_depth_to_points', 'kbe_shift_points',
    'kbe_spatial_filter', 'kbe_laplacian_valid', 'kbe_pconv_epilogue', 'kbe_prelu_mask', 'kbe_bias_act', 'kbe_upsample2x_act', 'kbe_frame_scratch_init_sets', 'kbe_video_handoff_status',
)

ABI_VERSION = 11
MAX_LANES = 8          # KBE_MAX_LANES
DEFAULT_LANES = 4      # streams the frame loop spreads consecutive frames over (env KBE_LANES)
FUSED_MAX_DENSITY = 20.0         # clouds up to this many points per pixel take the fused scatter by default (KBE_FUSED=auto).  Round 5: a tile's candidate list holds 2048
                                 # sub-blocks (512 before: the densest tiles' lists overflowed from 5 points per pixel on and those tiles scanned the cloud) and a record
                                 # that finds no room in LDS waits as its 4-byte point index (16-byte records before; four times the capacity) -- the route now grows
                                 # linearly to 20 points per pixel: 141 / 284 / 395 / 535 / 657 us per 1024^2 frame at 4 / 9 / 12 / 16 / 20 per pixel, 11.6 ms at 24
                                 # (profiles/r05_density_sweep.txt; the bucket route: 153 / 14 083 us at 4 / 9, the atomic kernels 2.1 / 5.2 / 20.5 ms at 4 / 9 / 24).
                                 # (4.5 in round 4, 1.5 before.)
FUSED_MAX_POINTS = 1 << 28       # the packed cloud's route addresses points by 32-bit byte offsets (KBE_FUSED_MAX_POINTS, kbe_tiles.h): larger clouds take the bucket route
FUSED_DENSE = 1.5                # "denser than the raster" from here on: delivered to host memory such a video takes three frames per launch on every lane (two until round 6)
FUSED_HOST_GROUP = 12  # frames per launch of the fused scatter when the frames are delivered to host memory (KBE_FILL_GROUP overrides): as many as a
                       # launch's 4 KB of kernel arguments hold.  A launch alone on a stream costs ~5 us besides its frames (ramp and tail: 8 / 12
                       # frames per launch 16.5 / 16.1 us per frame), and delivered to host memory -- the link binds -- a video runs at the same rate
                       # with 8 or 12 (17.4-17.6 k frames/s, --steps 75 16.1-16.3 k, --steps 20 13.8-13.9 k either way: tools/batches/gpu_r04_group12.sh)
DEFAULT_FILL_GROUP = 4 # frames a lane fills per launch when the table-driven fill is on (env KBE_FILL_GROUP, 1..4)
PROBE_MIN_FRAMES = 128 # delivered videos from this length on have their lanes measured (HipKernels.delivery_lanes); shorter ones take the a-priori estimate
DEFAULT_HOST_LANES = 2 # of those, the lanes used when the frames are delivered to pinned host memory AND the link binds (host_lanes below)


LINK_BYTES_PER_US = 53.0e3     # what the PCIe link moves for this loop (measured: 53-54.5 GB/s of the 63 GB/s Gen5 x16)


def lanes_for_delivery(lanes, render_us, frame_bytes):
    """Two lanes where the link binds (it needs 1.5 x longer per frame than the rendering), all lanes elsewhere."""
    link_us = frame_bytes / LINK_BYTES_PER_US
    return min(lanes, DEFAULT_HOST_LANES) if link_us > 1.5 * render_us else lanes


_LANE_STREAMS = {}


def lane_streams_of(device, n):
    """n streams for the lanes of the frame loop beside the caller's stream, one set per device for the life of the process: torch's own
    streams.  KBE_LANE_PRIORITY=1 (dev) makes them LOW-priority HIP streams: lane 0 -- the caller's stream -- renders a video's FIRST
    frame, which every byte of a short delivered video waits for, while lane 1 already renders the second group next to it; with the
    other lanes below it the first frame has the chip when both want it (tools/batches/gpu_r05_prio.sh, three processes each, k frames/s
    delivered: 20 frames 14.95 -> 15.02, 16 frames 14.28 -> 14.40; high-priority lanes 14.83 / 14.25; long videos and frames left in
    HBM: no difference).  NOT the default: with several processes on one GPU (tests/test_hip_parity.py: four ranks next to the test
    process) the low-priority queues are scheduled erratically -- passes of 3.5 to 24.6 ms where torch's streams give 7.0 +- 0.4."""
    device = torch.device(device)
    key = (device.index if device.index is not None else torch.cuda.current_device())
    have = _LANE_STREAMS.setdefault(key, [])
    while len(have) < n:
        stream = None
        if os.environ.get('KBE_LANE_PRIORITY', '0') == '1':
            try:
                hip = ctypes.CDLL('libamdhip64.so')
                least, greatest, handle = ctypes.c_int(), ctypes.c_int(), ctypes.c_void_p()
                with torch.cuda.device(key):
                    if hip.hipDeviceGetStreamPriorityRange(ctypes.byref(least), ctypes.byref(greatest)) == 0 and least.value > 0 and \
                            hip.hipStreamCreateWithPriority(ctypes.byref(handle), ctypes.c_uint(1), least) == 0 and handle.value:      # 1 = hipStreamNonBlocking
                        stream = torch.cuda.ExternalStream(handle.value, device=torch.device('cuda', key))
            except (OSError, AttributeError, RuntimeError):
                stream = None
        have.append(stream if stream is not None else torch.cuda.Stream(device=torch.device('cuda', key)))
    return have[:n]


def transfer_group(n_frames, lanes, launch_group, fast_ramp=False):
    """-(frames per transfer group) of a delivered video of n_frames on `lanes` lanes (kbe_render_video's batch < 0; the first groups
    ramp 1, 2, 4, ...: include/kbe.h).  Two lanes -- the link binds: groups of up to 32 frames (16 -> 32: 17.3 -> 17.5 k frames/s), a quarter of
    the video between 64 and 128 frames, 16 below.  More lanes -- the rendering binds, not the link (delivery_lanes): a
    lane then waits for nothing but its own last transfer, and what a video loses is its END -- the lanes' last groups leave one after
    the other when nothing is left to render, and groups of 32 deal the frames unevenly to four lanes: a transfer group is what ONE
    scatter launch renders (`launch_group`).  Measured (tools/batches/gpu_r05_dolly_batch.sh, profiles/r05_transfer_groups.txt): bench --dolly,
    256 frames, k frames/s delivered with groups of up to 32 / 16 / 12 / 8 frames 10.2 / 10.7 / 10.7 / 11.4 (left in HBM: 13.0);
    configs[4], 64 frames, groups of 32 / 2: 2.2 / 3.0 k."""
    if lanes > DEFAULT_HOST_LANES:
        return -max(1, int(launch_group))
    # (the ramp is never cut below 16: until late in round 5 a short video's cap was n / 4 alone -- "small enough for each lane to have two
    # groups of full size", a rule from the blit hand-off's days; with the SDMA engine a transfer fewer is worth more: tools/batches/gpu_r05_short_batch.sh,
    # k frames/s delivered with the old cap / 8 / 16: 16 frames 13.8-13.9 / 14.3 / 14.3, 20 frames 14.8 / 14.9 / 14.9, 30 frames 15.5 / 15.4 / 15.65; 40, 75: equal)
    return -max(1, min(32, max(n_frames // (2 * lanes), 16, (n_frames + 1) // 2 if fast_ramp else 0)))


def host_lanes(lanes, n_points, W, H, frame_bytes):
    """The a-priori estimate (HipKernels.delivery_lanes measures instead, once per cloud, when the video is long enough).
    Lanes of the frame loop when the frames go to pinned host memory (env KBE_HOST_LANES overrides).  Where the PCIe
    link binds, two lanes ping-pong best (one renders its next group while the other's leaves: 59.1 us per 1024^2 frame of
    the bench against 60.7 with four and 84.8 with three); where the rendering binds every lane helps (measured, 2 -> 4
    lanes: 512^2 25.5 -> 18.8 us, dolly 260 -> 155, raw cloud 64 -> 61, 2048^2 raw 305 -> 276, 2048^2 from 16.8 M points
    427 -> 386).  Which it is, from what is known before the first frame: the link needs frame_bytes / 53 GB/s per frame; a
    frame of an inpainted cloud (more points than pixels: few holes to fill) renders in about 13.5 us per million points
    + 12 us per megapixel with four lanes, and never in less than the ~14 us its four launches take; a cloud without
    appended points leaves holes whose fill dominates (rendering binds).  The link binds when it needs 1.5 x longer."""
    env = os.environ.get('KBE_HOST_LANES')
    if env:
        return min(lanes, max(1, int(env)))
    link_us = frame_bytes / 53.0e3
    render_us = max(14.0, 13.5e-6 * n_points + 12.0e-6 * W * H)
    link_bound = n_points > W * H and link_us > 1.5 * render_us
    return min(lanes, DEFAULT_HOST_LANES) if link_bound else lanes
_lib = None


class KbeError(RuntimeError):
    pass


def load():
    """Loads the HIP library once; fails loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KbeError('HIP extension not built: %s is missing. Run `python -c "import __graft_entry__ as g; g.build()"` '
                       '(hipcc --offload-arch=gfx950). There is no CPU fallback.' % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise KbeError('libkbe_hip.so does not export %s (stale build?)' % name)
        getattr(lib, name).restype = ctypes.c_int
    lib.kbe_last_error.restype = ctypes.c_char_p
    lib.kbe_frame_scratch_bytes.restype = ctypes.c_size_t
    lib.kbe_video_scratch_stride.restype = ctypes.c_size_t
    lib.kbe_video_stage_bytes.restype = ctypes.c_size_t
    lib.kbe_cloud_pack_bytes.restype = ctypes.c_size_t
    if lib.kbe_abi_version() != ABI_VERSION:
        raise KbeError('libkbe_hip.so ABI %d != expected %d' % (lib.kbe_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


_i, _d, _f, _z = ctypes.c_int, ctypes.c_double, ctypes.c_float, ctypes.c_size_t


def _ptr(t, dtype=torch.float32):
    if t is None:
        return ctypes.c_void_p(0)
    if not (t.is_cuda and t.dtype == dtype and t.is_contiguous()):
        raise KbeError('kernel arguments must be contiguous %s tensors on the GPU (got %s %s on %s)'
                       % (dtype, 'contiguous' if t.is_contiguous() else 'strided', t.dtype, t.device))
    if t.device.index != torch.cuda.current_device():
        raise KbeError('tensor on %s but the current device is cuda:%d: wrap the call in torch.cuda.device(%d) '
                       '(launches go to the current device\'s stream)' % (t.device, torch.cuda.current_device(), t.device.index))
    return ctypes.c_void_p(t.data_ptr())


def _f32c(t):
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    return t if t.is_contiguous() else t.contiguous()


def _shift(shift3):
    if shift3 is None:
        return None
    if torch.is_tensor(shift3):
        shift3 = shift3.detach().reshape(-1).tolist()     # 3 host floats (already fp32-representable)
    return (ctypes.c_float * 3)(*[float(v) for v in shift3])


def _stream():
    """The HIP stream launches go to: torch's current stream of the CURRENT device.  Every wrapper below checks that its
    tensors live on that device (`_ptr`), so a caller working on cuda:1 without torch.cuda.set_device(1) gets a KbeError
    instead of device-1 pointers launched on device 0's stream."""
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def fused_build_bits(video=False):
    """KBE_FUSED_CAP=lean|roomy (tests, measurements): the flag that forces one build of the fused route's tile launches
    (KBE_STAGE_FUSED_LEAN / _ROOMY, or the kbe_render_video form).  The environment is read HERE, per call; the library reads none."""
    cap = os.environ.get('KBE_FUSED_CAP')
    bits = {'lean': (1024, 2048), 'roomy': (2048, 4096)}.get(cap)
    return bits[1 if video else 0] if bits else 0


HANDOFF_DEFAULT = 'sdma'    # measured (profiles/r05_handoff_sdma.txt): 20 frames 14.1 k -> 14.5-14.8 k frames/s, 75 frames 16.4-16.5 -> 16.5-16.6 k, long videos equal (KBE_HANDOFF=sdma|blit)


def handoff_by_sdma():
    """Do a delivered video's frame groups leave through an SDMA engine (KBE_VIDEO_SDMA) or through hipMemcpyAsync (a blit kernel)?"""
    return os.environ.get('KBE_HANDOFF', HANDOFF_DEFAULT) == 'sdma'


def stride_of(K, state):
    return int(K.lib.kbe_video_scratch_stride(_i(state['W']), _i(state['H']), _i(state['N'])))


class HipKernels:
    """Tensor-level view of the C ABI.  One instance is shared by the package."""

    name = 'hip'

    def __init__(self):
        self.lib = load()
        self._tiled_scratch = {}      # (device, W, H) -> scratch of the tiled render_pointcloud

    def _check(self, rc, what):
        if rc != 0:
            raise KbeError('%s failed (%d): %s' % (what, rc, self.lib.kbe_last_error().decode()))

    def selftest_err(self, z, focal, baseline):
        z = _f32c(z).reshape(-1)
        fast, exact = torch.empty_like(z), torch.empty_like(z)
        self._check(self.lib.kbe_selftest_err(_ptr(z), _z(z.numel()), _d(float(focal)), _d(float(baseline)), _ptr(fast), _ptr(exact),
                                              _stream()), 'kbe_selftest_err')
        return fast, exact

    def selftest_division(self, num, den):
        num, den = _f32c(num).reshape(-1), _f32c(den).reshape(-1)
        fast, ieee = torch.empty_like(num), torch.empty_like(num)
        self._check(self.lib.kbe_selftest_division(_ptr(num), _ptr(den), _z(num.numel()), _ptr(fast), _ptr(ieee), _stream()), 'kbe_selftest_division')
        return fast, ieee

    # -- render_pointcloud and its stages ------------------------------------------------
    def zsplat(self, points, W, H, focal, baseline, shift3=None, want_winner=False):
        points = _f32c(points)
        B, _, N = points.shape
        zkeys = torch.empty(B, 1, H, W, dtype=torch.int32, device=points.device)
        self._check(self.lib.kbe_zkeys_clear(_ptr(zkeys, torch.int32), _z(zkeys.numel()), _stream()), 'kbe_zkeys_clear')
        winner = torch.empty(B, N, dtype=torch.int32, device=points.device) if want_winner else None
        self._check(self.lib.kbe_zsplat(_ptr(points), _i(B), _i(N), _i(W), _i(H), _d(float(focal)), _d(float(baseline)),
                                        _shift(shift3), _ptr(zkeys, torch.int32), _ptr(winner, torch.int32), _stream()),
                    'kbe_zsplat')
        return zkeys, winner

    def zkeys_decode(self, zkeys):
        zee = torch.empty(zkeys.shape, dtype=torch.float32, device=zkeys.device)
        self._check(self.lib.kbe_zkeys_decode(_ptr(zkeys, torch.int32), _z(zkeys.numel()), _ptr(zee), _stream()),
                    'kbe_zkeys_decode')
        return zee

    def degrid(self, zkeys=None, zee=None):
        src = zkeys if zkeys is not None else zee
        B, _, H, W = src.shape
        out = torch.empty(B, 1, H, W, dtype=torch.float32, device=src.device)
        self._check(self.lib.kbe_degrid(_ptr(zkeys, torch.int32), _ptr(None if zee is None else _f32c(zee)), _i(B), _i(W),
                                        _i(H), _ptr(out), _stream()), 'kbe_degrid')
        return out

    def degrid_serial(self, zkeys=None, zee=None):
        """The serial (index-order, in-place) schedule of the same kernel: what the reference-run golden vectors hold.
        Proof-of-fidelity entry for the tests; the product path uses :meth:`degrid`'s out-of-place schedule."""
        src = zkeys if zkeys is not None else zee
        B, _, H, W = src.shape
        out = torch.empty(B, 1, H, W, dtype=torch.float32, device=src.device)
        self._check(self.lib.kbe_degrid_serial(_ptr(zkeys, torch.int32), _ptr(None if zee is None else _f32c(zee)), _i(B), _i(W),
                                               _i(H), _ptr(out), _stream()), 'kbe_degrid_serial')
        return out

    def accumulate(self, points, data, zee, focal, baseline, shift3=None):
        points, data = _f32c(points), _f32c(data)
        B, C, N = data.shape
        _, _, H, W = zee.shape
        acc = torch.zeros(B, C + 1, H, W, dtype=torch.float32, device=points.device)
        self._check(self.lib.kbe_accumulate(_ptr(points), _ptr(data), _i(B), _i(N), _i(C), _ptr(_f32c(zee)), _i(W), _i(H),
                                            _d(float(focal)), _d(float(baseline)), _shift(shift3), _ptr(acc), _stream()),
                    'kbe_accumulate')
        return acc

    def normalize(self, acc):
        B, C1, H, W = acc.shape
        render = torch.empty(B, C1 - 1, H, W, dtype=torch.float32, device=acc.device)
        existing = torch.empty(B, 1, H, W, dtype=torch.float32, device=acc.device)
        self._check(self.lib.kbe_normalize(_ptr(_f32c(acc)), _i(B), _i(C1 - 1), _i(W), _i(H), _ptr(render), _ptr(existing),
                                           _stream()), 'kbe_normalize')
        return render, existing

    def render_pointcloud(self, points, data, W, H, focal, baseline, tiled=None):
        """common.py:428-686 -> (render [B,C,H,W], existing [B,1,H,W]).  ``tiled`` (default: env KBE_RENDER_TILED, on):
        one sample at a time through the tile machinery of the frame loop (no float atomics; 20x faster at 68
        channels); off: the stage-by-stage global-atomic kernels (kept as the independent cross-check)."""
        points, data = _f32c(points), _f32c(data)
        B, C, N = data.shape
        dev = points.device
        if tiled is None:
            # the tile renderer's size limits (include/kbe.h); beyond them the stage-by-stage HIP kernels take over
            tiled = (os.environ.get('KBE_RENDER_TILED', '1') != '0' and N <= (1 << 30) and int(W) * int(H) <= (1 << 30)
                     and int(W) < (1 << 24) and int(H) < (1 << 24))
        if tiled:
            W, H = int(W), int(H)
            key = (dev, W, H)
            scratch = self._tiled_scratch.get(key)
            if scratch is None:
                scratch = torch.empty(int(self.lib.kbe_frame_scratch_bytes(_i(W), _i(H), _i(0))), dtype=torch.uint8, device=dev)
                self._check(self.lib.kbe_frame_scratch_init(_ptr(scratch, torch.uint8), _i(W), _i(H), _stream()), 'kbe_frame_scratch_init')
                self._tiled_scratch = {key: scratch}           # one size at a time (200 MB at 1024^2)
            render = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
            existing = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
            for b in range(B):
                self._check(self.lib.kbe_render_pointcloud_tiled(_ptr(points[b]), _ptr(data[b]), _i(N), _i(C), _i(W), _i(H),
                                                                 _d(float(focal)), _d(float(baseline)), None, _ptr(scratch, torch.uint8),
                                                                 _ptr(render[b]), _ptr(existing[b]), _stream()),
                            'kbe_render_pointcloud_tiled')
            return render, existing
        zkeys = torch.empty(B * H * W, dtype=torch.int32, device=dev)
        zee = torch.empty(B * H * W, dtype=torch.float32, device=dev)
        acc = torch.empty(B * (C + 1) * H * W, dtype=torch.float32, device=dev)
        render = torch.empty(B, C, H, W, dtype=torch.float32, device=dev)
        existing = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev)
        self._check(self.lib.kbe_render_pointcloud(_ptr(points), _ptr(data), _i(B), _i(N), _i(C), _i(W), _i(H),
                                                   _d(float(focal)), _d(float(baseline)), _ptr(zkeys, torch.int32),
                                                   _ptr(zee), _ptr(acc), _ptr(render), _ptr(existing), _stream()),
                    'kbe_render_pointcloud')
        return render, existing

    # -- fill ---------------------------------------------------------------------------
    def fill_disocclusion(self, x, depth):
        x, depth = _f32c(x), _f32c(depth)
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        self._check(self.lib.kbe_fill_disocclusion(_ptr(x), _ptr(depth), _i(B), _i(C), _i(W), _i(H), _ptr(out), _stream()),
                    'kbe_fill_disocclusion')
        return out

    # -- the frame loop on the resident cloud ---------------------------------------------
    def prepare_cloud(self, points, image, depth, W, H, focal=None, raster=None, near_depth=None):
        """Makes tensorInpaPoints/Image/Depth resident for the frame loop: contiguous fp32 views (used in
        place, no repacking) plus the per-view scratch (z-buffer, tile buckets, hole list), initialised
        once.  Returns the state render_frame consumes."""
        W, H = int(W), int(H)
        N = points.shape[-1]
        dev = points.device
        state = {'points': _f32c(points).reshape(3, N), 'image': _f32c(image).reshape(3, N), 'depth': _f32c(depth).reshape(N),
                 'N': N, 'W': W, 'H': H, 'frame': torch.empty(H, W, 3, dtype=torch.uint8, device=dev),
                 # layout hint: process_kenburns' cloud starts with the W x H image raster (common.py:176-179)
                 'raster_w': W if N >= W * H else 0, 'raster_n': W * H if N >= W * H else 0}
        if raster is not None:          # (width, count): the cloud starts with a row-major raster of another shape
            state['raster_w'], state['raster_n'] = int(raster[0]), int(raster[1])
        if os.environ.get('KBE_NO_RASTER_HINT'):
            state['raster_w'] = state['raster_n'] = 0
        # one scratch per lane of the frame loop (render_video renders consecutive frames on `lanes` streams);
        # render_frame uses lane 0's
        lanes = max(1, min(MAX_LANES, int(os.environ.get('KBE_LANES', DEFAULT_LANES))))
        stride = int(self.lib.kbe_video_scratch_stride(_i(W), _i(H), _i(N)))
        state['lanes'] = lanes
        state['scratch'] = torch.empty(lanes * stride, dtype=torch.uint8, device=dev)
        self._check(self.lib.kbe_frame_scratch_init_sets(ctypes.c_void_p(state['scratch'].data_ptr()), _z(stride), _i(lanes), _i(W), _i(H), _stream()),
                    'kbe_frame_scratch_init_sets')
        # Two routes for the scatter of a frame, same results (tests/test_hip_parity.py::test_fused_scatter_equals_the_bucket_path):
        #   fused   k_place -> k_frame on the packed cloud (kbe_cloud_pack, once per cloud): every point projected once, its
        #           12-byte placement stored in place, a tile pulls the sub-blocks listed for it, z-tile in LDS; no per-point
        #           global atomic, 1.4x the algorithmic HBM bytes;
        #   bucket  k_project -> k_tiles: 16-byte records appended to per-tile buckets, z-buffer in HBM: 2.0x the bytes.
        # Measured, us per frame left in HBM, bucket / fused: 1024^2 inpainted cloud 29.4 / 25.3, raw 37.4 / 35.5, 2048^2 raw
        # 144 / 122, 512^2 9.3 / 9.1, 256^2 4.3 / 4.5 -- but clouds much denser than the raster pile more records on a tile
        # than its LDS holds and the fused route's further rounds go through HBM: 2048^2 from 16.8 M points 375 / 403, and a
        # dolly zoom-out (the image shrinks, the density grows along the video) 97 / 151.  KBE_FUSED = auto (default: fused
        # unless the cloud has more than 1.5 points per pixel; render_video also looks at the camera path) | 1 | 0.
        mode = os.environ.get('KBE_FUSED', 'auto')
        state['fused'] = ((N <= FUSED_MAX_DENSITY * W * H) if mode == 'auto' else mode not in ('0', 'generic')) and N <= FUSED_MAX_POINTS
        # ... and a third, on request only (KBE_FUSED=generic): the reference's own decomposition (z-splat, degrid, accumulate with global
        # atomics, normalise, fill), one frame at a time -- the independent cross-check of the two tile routes as a whole video loop.  (For a
        # day of round 5 clouds beyond 6.5 points per pixel took it, as the lesser evil next to the tile routes' cliff; with the cliff gone it
        # is 15-35 x slower than the fused route at every density measured.)
        state['generic'] = mode == 'generic'
        state['cloud_focal'] = float(focal) if focal else 512.0
        # the depth of the nearest point (the reference's objectDepthrange[0], common.py:88, when the caller has it; else found
        # on first need): decides how many consecutive frames of a video share candidate lists (include/kbe.h: near_depth)
        state['near_depth'] = None if near_depth is None else (float(near_depth) if 0.0 < float(near_depth) < 1.0e30 else 0.0)     # (nan, inf, <= 0: unknown)
        if state['fused']:
            self._pack(state)
        return state

    @staticmethod
    def near_depth(state):
        """The nearest depth of the cloud, for kbe_render_video / kbe_render_frame_group_ahead's `near_depth` (KBE_SHARE_LISTS=0: 0.0,
        every frame keeps candidate lists of its own).  Found with one reduction and one host synchronisation where the caller of
        prepare_cloud did not pass it."""
        if os.environ.get('KBE_SHARE_LISTS') == '0':
            return 0.0
        if state.get('near_depth') is None:
            z = state['points'][2]
            z = z[z > 0]
            near = float(z.min()) if z.numel() else 0.0
            state['near_depth'] = near if 0.0 < near < 1.0e30 else 0.0
        return state['near_depth']

    def _pack(self, state):
        """kbe_cloud_pack, once per cloud (on first use of the fused route)."""
        if 'packed' not in state:
            N, W, H = state['N'], state['W'], state['H']
            state['packed'] = torch.empty(int(self.lib.kbe_cloud_pack_bytes(_i(N))), dtype=torch.uint8, device=state['points'].device)
            self._check(self.lib.kbe_cloud_pack(_ptr(state['points']), _ptr(state['image']), _ptr(state['depth']), _i(N), _i(W), _i(H),
                                                _d(state['cloud_focal']), _i(state['raster_w']), _i(state['raster_n']),
                                                _ptr(state['packed'], torch.uint8), _stream()), 'kbe_cloud_pack')

    def render_frame(self, state, shift3, focal, baseline, render_f32=None, existing_f32=None, zee_f32=None,
                     zee_pre_f32=None, out=None, stages=7, fill_rect=None, fused=None, parity=-1):
        """One frame of common.py:238-255 (shift -> render -> fill -> uint8) -> uint8 [H,W,3] on the device.
        fill_rect = (x0, y0, x1, y1): only holes inside are filled (see include/kbe.h).  ``fused`` (default: the
        state's route): the one-launch scatter on the packed cloud; False: the bucket path (k_project + k_tiles).
        ``stages``: bit 1 = projection launch (bucket path only), 2 = scatter / tile launch, 4 = hole fill (+ flags).
        ``parity`` (fused route, include/kbe.h): -1 = a frame on its own; 0, 1, 0, ... for consecutive frames on one scratch."""
        frame = state['frame'] if out is None else out
        rect = None if fill_rect is None else (ctypes.c_int * 4)(*[int(v) for v in fill_rect])
        if state.get('fused') if fused is None else fused:
            self._pack(state)
            self._check(self.lib.kbe_render_frame_fused(_ptr(state['packed'], torch.uint8), _i(state['N']), _d(state['cloud_focal']),
                                                        _i(state['W']), _i(state['H']), _d(float(focal)), _d(float(baseline)),
                                                        _shift(shift3), _ptr(state['scratch'], torch.uint8), _ptr(frame, torch.uint8),
                                                        _ptr(render_f32), _ptr(existing_f32), _ptr(zee_f32), _ptr(zee_pre_f32),
                                                        _i((int(stages) & ~1) | fused_build_bits()), rect, _i(int(parity)), _stream()), 'kbe_render_frame_fused')
            return frame
        self._check(self.lib.kbe_render_frame_stages(_ptr(state['points']), _ptr(state['image']), _ptr(state['depth']),
                                                     _i(state['N']), _i(state['W']), _i(state['H']), _d(float(focal)),
                                                     _d(float(baseline)), _shift(shift3), _ptr(state['scratch'], torch.uint8),
                                                     _ptr(frame, torch.uint8), _ptr(render_f32), _ptr(existing_f32),
                                                     _ptr(zee_f32), _ptr(zee_pre_f32), _i(int(stages)), rect, _i(state['raster_w']),
                                                     _i(state['raster_n']), _stream()),
                    'kbe_render_frame')
        return frame

    def group_scratch(self, state, sets):
        """`sets` initialised scratch sets for launches that take several frames (kbe_render_frame_group, KBE_VIDEO_FILL_GROUP),
        allocated on first use: (tensor, stride in bytes)."""
        stride = self.scratch_stride(state)
        if 'scratch_groups' not in state or state['scratch_groups'].numel() < sets * stride:
            state['scratch_groups'] = torch.empty(sets * stride, dtype=torch.uint8, device=state['points'].device)
            self._check(self.lib.kbe_frame_scratch_init_sets(ctypes.c_void_p(state['scratch_groups'].data_ptr()), _z(stride), _i(sets), _i(state['W']), _i(state['H']),
                                                             _stream()), 'kbe_frame_scratch_init_sets')
        return state['scratch_groups'], stride

    def render_frame_group(self, state, cameras, baseline, out, stages=7, zbuf_flags=None, fill_rect=None):
        """kbe_render_frame_group: the launches of 1..4 frames [(focal, shift3)] of the bucket route, every launch taking all of
        them; out: uint8 [n,H,W,3] on the device.  Each frame uses a scratch set of its own (state['scratch_groups'])."""
        n = len(cameras)
        scratch, stride = self.group_scratch(state, max(n, 4))
        focals = (ctypes.c_double * n)(*[float(c[0]) for c in cameras])
        shifts = (ctypes.c_float * (3 * n))(*[float(v) for c in cameras for v in c[1]])
        sets = (ctypes.c_void_p * n)(*[scratch.data_ptr() + k * stride for k in range(n)])
        frames = (ctypes.c_void_p * n)(*[out[k].data_ptr() for k in range(n)])
        zf = None if zbuf_flags is None else (ctypes.c_int * n)(*[int(v) for v in zbuf_flags])
        rect = None if fill_rect is None else (ctypes.c_int * 4)(*[int(v) for v in fill_rect])
        self._check(self.lib.kbe_render_frame_group(_ptr(state['points']), _ptr(state['image']), _ptr(state['depth']), _i(state['N']), _i(state['W']),
                                                    _i(state['H']), _d(float(baseline)), _i(n), focals, shifts, sets, frames, zf, _i(int(stages)), rect,
                                                    _i(state['raster_w']), _i(state['raster_n']), _stream()), 'kbe_render_frame_group')
        return out

    def render_frame_group_fused(self, state, cameras, baseline, out, stages=6, parities=None, fill_rect=None):
        """kbe_render_frame_group_fused: the same on the packed cloud (k_place + k_frame + fill, each taking all the frames)."""
        n = len(cameras)
        self._pack(state)
        scratch, stride = self.group_scratch(state, max(n, 4) if 'scratch_groups' not in state else max(n, state['scratch_groups'].numel() // stride_of(self, state)))
        focals = (ctypes.c_double * n)(*[float(c[0]) for c in cameras])
        shifts = (ctypes.c_float * (3 * n))(*[float(v) for c in cameras for v in c[1]])
        sets = (ctypes.c_void_p * n)(*[scratch.data_ptr() + k * stride for k in range(n)])
        frames = (ctypes.c_void_p * n)(*[out[k].data_ptr() for k in range(n)])
        par = None if parities is None else (ctypes.c_int * n)(*[int(v) for v in parities])
        rect = None if fill_rect is None else (ctypes.c_int * 4)(*[int(v) for v in fill_rect])
        self._check(self.lib.kbe_render_frame_group_fused(_ptr(state['packed'], torch.uint8), _i(state['N']), _d(state['cloud_focal']), _i(state['W']),
                                                          _i(state['H']), _d(float(baseline)), _i(n), focals, shifts, sets, frames, par, _i(int(stages) | fused_build_bits()), rect,
                                                          _stream()), 'kbe_render_frame_group_fused')
        return out

    def render_frame_group_ahead(self, state, cameras, baseline, out, turn, placed, next_cameras=None, stages=6, fill_rect=None, next_turn=None):
        """kbe_render_frame_group_ahead: frame k on scratch set k of the group scratch, on the set's turn `turn` (an int for all, or
        one per frame); the tile launch makes the placements of `next_cameras` when given (frame k on set k, turn `next_turn`:
        default turn + 1)."""
        n = len(cameras)
        m = len(next_cameras) if next_cameras else 0
        self._pack(state)
        scratch, stride = self.group_scratch(state, max(n, m, 4) if 'scratch_groups' not in state else max(n, m, state['scratch_groups'].numel() // stride_of(self, state)))

        def arrays(cams, k):
            return ((ctypes.c_double * k)(*[float(c[0]) for c in cams]), (ctypes.c_float * (3 * k))(*[float(v) for c in cams for v in c[1]]),
                    (ctypes.c_void_p * k)(*[scratch.data_ptr() + j * stride for j in range(k)]))
        focals, shifts, sets = arrays(cameras, n)
        nf, ns, nsets = arrays(next_cameras, m) if m else (None, None, None)
        frames = (ctypes.c_void_p * n)(*[out[k].data_ptr() for k in range(n)])
        turn = [int(turn)] * n if isinstance(turn, int) else [int(t) for t in turn]
        if next_turn is None:
            next_turn = [(turn[k] if k < n else turn[0]) + 1 for k in range(m)]
        turns = (ctypes.c_int * n)(*turn)
        nturns = (ctypes.c_int * m)(*[int(t) for t in next_turn]) if m else None
        rect = None if fill_rect is None else (ctypes.c_int * 4)(*[int(v) for v in fill_rect])
        self._check(self.lib.kbe_render_frame_group_ahead(_ptr(state['packed'], torch.uint8), _i(state['N']), _d(state['cloud_focal']), _i(state['W']),
                                                          _i(state['H']), _d(float(baseline)), _i(n), focals, shifts, sets, frames, turns, _i(1 if placed else 0),
                                                          _i(m), nf, ns, nsets, nturns, _i(int(stages) | fused_build_bits()), rect, _d(self.near_depth(state)), _stream()), 'kbe_render_frame_group_ahead')
        return out

    def prepared_group_ahead(self, state, cameras, baseline, out, next_cameras, stages=2):
        """A closure that makes the launch of :meth:`render_frame_group_ahead` for a FIXED group -- frame k on scratch set k, the next
        group the same sets -- with every argument array built once: ``launch(turn, placed, place_next=True)`` only writes the turn
        numbers.  For timing loops: building the ctypes arrays of two groups of eight cameras costs the host about as long as the
        GPU works on the launch, and a measurement through the ordinary wrapper then reads the host's speed."""
        n, m = len(cameras), len(next_cameras)
        self._pack(state)
        scratch, stride = self.group_scratch(state, max(n, m, 4) if 'scratch_groups' not in state else max(n, m, state['scratch_groups'].numel() // stride_of(self, state)))

        def arrays(cams, k):
            return ((ctypes.c_double * k)(*[float(c[0]) for c in cams]), (ctypes.c_float * (3 * k))(*[float(v) for c in cams for v in c[1]]),
                    (ctypes.c_void_p * k)(*[scratch.data_ptr() + j * stride for j in range(k)]))
        focals, shifts, sets = arrays(cameras, n)
        nf, ns, nsets = arrays(next_cameras, m)
        frames = (ctypes.c_void_p * n)(*[out[k].data_ptr() for k in range(n)])
        turns, nturns = (ctypes.c_int * n)(), (ctypes.c_int * m)()
        fixed = (_ptr(state['packed'], torch.uint8), _i(state['N']), _d(state['cloud_focal']), _i(state['W']), _i(state['H']), _d(float(baseline)), _i(n),
                 focals, shifts, sets, frames, turns)
        tail = (nf, ns, nsets, nturns, _i(int(stages) | fused_build_bits()), None, _d(self.near_depth(state)))
        fn, check, zero = self.lib.kbe_render_frame_group_ahead, self._check, _i(0)
        keep = (scratch, out)          # (the arrays hold raw addresses)

        def launch(turn, placed, place_next=True):
            for k in range(n):
                turns[k] = turn
            for k in range(m):
                nturns[k] = turn + 1
            check(fn(*fixed, _i(1 if placed else 0), _i(m) if place_next else zero, *tail, _stream()), 'kbe_render_frame_group_ahead')
        launch.keep = keep
        return launch

    def scratch_stride(self, state):
        """Bytes between two scratch sets of the cloud's frame size (kbe_video_scratch_stride), asked once per cloud."""
        if 'scratch_stride' not in state:
            state['scratch_stride'] = int(self.lib.kbe_video_scratch_stride(_i(state['W']), _i(state['H']), _i(state['N'])))
        return state['scratch_stride']

    def scratch_set_budget(self, state):
        """How many scratch sets (kbe_video_scratch_stride bytes each) a video loop may hold: KBE_SCRATCH_BUDGET_MB, or half of the
        device memory that is free now plus what the cloud's sets already hold; at least one per lane."""
        stride = self.scratch_stride(state)
        env = os.environ.get('KBE_SCRATCH_BUDGET_MB')
        if env:
            budget = float(env) * 1e6
        else:
            free, _ = torch.cuda.mem_get_info(state['points'].device)
            held = state['scratch_groups'].numel() if 'scratch_groups' in state else 0
            budget = 0.5 * (free + held)
        return max(state['lanes'], int(budget // stride))

    def video_launch_shape(self, state, cameras, batch, to_host=False, max_group=None):
        """(flags of kbe_render_video, frames per launch, fused route?) for a video of `cameras`.
        KBE_VIDEO_FILL_DIST, the table-driven hole fill: for videos whose frames have hundreds of thousands of holes -- a
        cloud without appended points (no inpainting) seen by a camera that zooms out (a dolly zoom lowers the focal length:
        the image shrinks into an empty border).  Measured, us per frame without / with: dolly 300 / 129 at 1024^2, 77 / 63
        at 512^2 -- but a raw cloud on the ordinary camera path 35.2 / 37.8, 2048^2 139 / 151 (two more launches per frame
        that find few holes).  KBE_FILL_DIST=1 / 0 forces it on / off.
        KBE_VIDEO_FILL_GROUP(n): a lane renders n frames into n scratch sets, every launch taking all n.  A launch on its
        own is bound by its ramp and its tail as much as by its work (the fused scatter of a 1024^2 frame: 35 us alone, 27 /
        23 per frame with 2 / 4 frames per launch), so frames left in HBM take 27.5 / 25.3 / 26.7 us with n = 1 / 2 / 4 on
        four lanes (the lanes fill the same gaps), and where the PCIe link binds (frames delivered to host memory: 59 us
        per frame whatever n) n = 4 leaves the most of the chip idle.  Small frames are bound by their launches: 4 up to
        576^2.  KBE_FILL_GROUP overrides.
        The route: the state's (prepare_cloud)."""
        W, H = state['W'], state['H']
        mode = os.environ.get('KBE_FILL_DIST', 'auto')
        zooms_out = len(cameras) > 0 and min(float(c[0]) for c in cameras) < 0.9 * state['cloud_focal']
        flags = int(state['N'] <= W * H and zooms_out) if mode == 'auto' else int(mode != '0')
        # (until round 5 a zoom-out took the bucket route: the density of the points on the shrinking image grows along the video, the fused
        # route's candidate lists of 512 sub-blocks overflowed and it lost 147 : 97 us per frame.  With lists of 2048 it wins: 82 against
        # 90 us per 1024^2 dolly frame left in HBM, 21.8 against 23.5 at 512^2 -- profiles/r05_dolly_routes.txt.  KBE_FUSED=0 still forces the other.)
        fused = bool(state.get('fused'))
        if os.environ.get('KBE_FILL_GROUP'):
            group = max(1, min(12 if fused else 4, int(os.environ['KBE_FILL_GROUP'])))
        elif flags:
            # the table-driven fill: launches bound by their own chains of look-ups -- four frames per fill launch; the fused scatter in
            # front of them takes eight (measured, us per dolly frame left in HBM / k frames/s delivered with 4 / 8 / 12: 82.1 / 80.4 / 79.7, 9.6 / 9.9 / 9.8)
            group = 8 if fused else DEFAULT_FILL_GROUP
        elif fused:
            # eight where the link binds (the rendering then only has to stay out of the transfers' way: the fewer, larger
            # launches the better), four for frames left in HBM on four lanes -- since a group's tile launch also makes the next
            # group's placements (one scatter launch per group) four frames per launch are as good as or better than two at every
            # size (measured, us per frame left in HBM with 2 / 4 / 8 frames per launch: 640^2 12.2 / 11.3 / 11.4, 896^2 20.3 / 20.9 / 21.3,
            # 1024^2 25.5 / 24.9 / 25.0, 1280^2 41.4 / 40.5 / 41.2, 1536^2 59.3 / 58.4 / 58.5; with a placement launch per group, round
            # 3's first half: 1024^2 25.3 / 26.7; the bucket route: 13.7, 18.8, 25.5, 29.4, 51.7, 72.8)
            group = FUSED_HOST_GROUP if to_host else 4
            if state['N'] > FUSED_DENSE * W * H:              # (left in HBM too since round 6: four per launch was the slowest of 2 / 3 / 4 there)
                # a cloud much denser than the raster is bound by its rendering, not by the link: few frames per launch on every lane.  Round 4
                # (blit hand-off: long launches next to another lane's copy kernel slowed each other), 16.8 M points at 2048^2, us per
                # delivered frame with 8 / 4 / 2 frames per launch on four lanes: 446 / 422 / 395.  Round 6, SDMA hand-off, frames/s delivered /
                # left in HBM with 2 / 3 / 4 / 6 / 8: 2 948 / 3 234, 2 941 / 3 246, 2 707 / 3 141, 2 598 / 2 999, 2 390 / 2 974 -- and the launch
                # alone on a stream 289.5 / 273.8 / 269.1 / 265.6 / 263.7 us per frame (its tail amortised): three is as fast as two for the
                # video and 5 % faster per launch (tools/batches/gpu_r06_config4_groups.sh, profiles/r06_config4_groups.txt)
                group = 3
        else:
            # the bucket route (measured, us per frame with 1 / 2 / 4 frames per launch: 256^2 13.3 / 9.8 / 6.2, 512^2 13.7 / 9.8 /
            # 8.6, 640^2 15.1 / 12.8 / 13.1, 768^2 19.1 / 16.4 / 17.4, 896^2 24.8 / 23.8 / 24.3, 1024^2 28.9 / 30.6 / 30.9)
            group = 4 if W * H <= 576 * 576 else (2 if W * H <= 900 * 900 else 1)
        if not (batch is None or batch <= 0):
            group = 1
        if not fused:
            group = min(group, 4)
        if max_group is not None:
            group = max(1, min(group, int(max_group)))
        return flags | (((group - 1) << 1) if group <= 4 else ((group - 1) << 5)), group, fused

    @staticmethod
    def zooms_out(state, cameras):
        return bool(cameras) and min(float(c[0]) for c in cameras) < 0.9 * state['cloud_focal']

    def delivery_lanes(self, state, cameras, baseline, crop=None):
        """Lanes of the frame loop when the frames go to pinned host memory: two where the link binds, all where the rendering
        does (`host_lanes` has the measurements).  Which it is depends on the cloud, the camera path and the route, so it is
        MEASURED once per cloud, camera-path kind and crop -- twelve of the video's frames rendered into HBM on all lanes, timed
        with two events (a few hundred microseconds, and ONE host synchronisation: the first render_video of a cloud is not
        enqueue-only) -- and kept in the cloud's state; videos shorter than PROBE_MIN_FRAMES (the product's 75 frames: the probe would
        be a third of their work), and the first lookup without a camera path, take the a-priori estimate.  A rank that received the cloud by broadcast takes what rank 0 measured (`delivery_lanes_hint`:
        sharding.measure_delivery_lanes) and never probes.  Env KBE_HOST_LANES overrides."""
        lanes, W, H = state['lanes'], state['W'], state['H']
        env = os.environ.get('KBE_HOST_LANES')
        if env:
            return min(lanes, max(1, int(env)))
        zoom = self.zooms_out(state, cameras)                  # a zoom-out renders differently: its own entry
        hint = state.get('delivery_lanes_hint', {})
        if zoom in hint:                                       # measured on rank 0, broadcast with the cloud (sharding.py)
            return min(lanes, max(1, int(hint[zoom])))
        key = (zoom, None if crop is None else (int(crop[0]), int(crop[1])))
        cache = state.setdefault('delivery_lanes', {})
        if key in cache:
            return cache[key]
        # (a video shorter than PROBE_MIN_FRAMES takes the a-priori estimate: the probe renders 24 frames and synchronises the host --
        # 0.5 ms of the 1.4 ms a 64-frame 512^2 video takes, paid by every Pipeline call, each with a cloud of its own: ADVICE r3)
        if len(cameras) < PROBE_MIN_FRAMES or lanes == 1:
            return host_lanes(lanes, state['N'], W, H, 3 * W * H)
        probe = cameras[::max(1, len(cameras) // 12)][:12]
        out = torch.empty(len(probe), H, W, 3, dtype=torch.uint8, device=state['points'].device)
        self.render_video(state, probe, baseline, crop=crop, host_out=out)                          # warm-up: scratch, streams, the packed cloud
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self.render_video(state, probe, baseline, crop=crop, host_out=out)
        e1.record()
        e1.synchronize()
        render_us = e0.elapsed_time(e1) * 1e3 / len(probe)
        cache[key] = lanes_for_delivery(lanes, render_us, 3 * W * H)       # a cropped frame is resized back to W x H (common.py:257)
        state['delivery_probe_us'] = render_us
        return cache[key]

    def _render_video_generic(self, state, cameras, baseline, crop, host_out):
        """The frame loop on the stage-by-stage kernels (kbe_shift_points, kbe_render_pointcloud's z-splat / degrid / accumulate /
        normalise with global atomics, kbe_fill_disocclusion, kbe_frame_u8, kbe_crop_resize_u8) -- the reference's own decomposition
        of common.py:238-257 (KBE_FUSED=generic: a cross-check of the tile routes, not a product route).  Frame by frame from Python:
        at >= 1 ms of GPU work per frame the loop's own cost does not matter."""
        n, W, H = len(cameras), state['W'], state['H']
        dev = state['points'].device
        if 'generic_data' not in state:
            state['generic_data'] = torch.cat([state['image'], state['depth'].reshape(1, -1)], 0).unsqueeze(0).contiguous()      # [1,4,N]: image; depth
        pts0, data = state['points'].unsqueeze(0), state['generic_data']
        out = host_out if host_out.is_cuda else torch.empty(n, H, W, 3, dtype=torch.uint8, device=dev)
        for i, (focal, shift3) in enumerate(cameras):
            pts = self.shift_points(pts0, shift3)                                                           # common.py:104-109
            render, existing = self.render_pointcloud(pts, data, W, H, focal, baseline, tiled=False)        # :428-686
            render = self.fill_disocclusion(render, render[:, 3:4] * (existing > 0.0).float())              # :253
            frame = self.frame_u8(render[:, 0:3])                                                           # :255
            out[i].copy_(frame if crop is None else self.crop_resize_u8(frame, int(crop[0]), int(crop[1])))  # :256-257
        if not host_out.is_cuda:
            host_out[:n].copy_(out, non_blocking=True)
        return host_out

    def render_video(self, state, cameras, baseline, crop=None, host_out=None, overlap=True, batch=None):
        """The frame loop for a list of (focal, shift3) cameras, enqueued natively; returns the pinned host
        tensor [n,H,W,3] the frames land in (valid after the current stream is synchronised).  ``host_out`` may
        also be a DEVICE tensor: the frames then stay in HBM (the last kernel of every frame stores straight into
        it, no transfer).  ``batch`` selects the hand-off to host memory (None: groups of up to 32 frames per
        transfer on KBE_HOST_LANES lanes; see include/kbe.h); ``overlap`` only matters for ``batch`` > 0."""
        n, W, H = len(cameras), state['W'], state['H']
        dev = state['points'].device
        if host_out is None:
            host_out = torch.empty(n, H, W, 3, dtype=torch.uint8, pin_memory=True)
        assert host_out.dtype == torch.uint8 and host_out.is_contiguous() and host_out.numel() >= n * H * W * 3
        if state.get('generic'):
            if not host_out.is_cuda and not host_out.is_pinned():
                raise KbeError('render_video: host_out must be pinned host memory (or a device tensor)')
            return self._render_video_generic(state, cameras, baseline, crop, host_out)
        lanes = state['lanes']
        if host_out.is_cuda:
            batch, overlap = 0, False
        else:
            if not host_out.is_pinned():
                raise KbeError('render_video: host_out must be pinned host memory (or a device tensor)')
            # The hand-off (include/kbe.h): < 0 = groups of -batch frames per lane, one hipMemcpyAsync each, the lanes
            # taking turns on the link (default); 0 = per frame by a copy kernel; > 0 = round 1's staged ring.
            lanes = self.delivery_lanes(state, cameras, baseline, crop)
            if batch is None:
                env = os.environ.get('KBE_DELIVERY_BATCH', '0')
                try:
                    batch = int(env)
                except ValueError:
                    raise KbeError('KBE_DELIVERY_BATCH=%r is not an integer (frames per transfer: < 0 groups per lane, > 0 staged ring)' % env)
                # how many frames a transfer group may hold: transfer_group (the first groups ramp 1, 2, 4, 8, 16: include/kbe.h)
                # (KBE_RAMP=fast: groups of 1, 3, 7, 15, 31, ... frames, capped at half the video -- two transfers fewer than 1, 2, 4, 8, ...
                # for a 20- or a 75-frame video.  Measured, round 4 (profiles/r04_short_videos.txt): no gain -- 20 frames 14.05 against
                # 13.93 k frames/s, 75 frames 16.05 against 16.21 k: the larger groups render next to the other lane's transfer, whose
                # blit kernel's PCIe-bound stores slow them.  Not the default.)
                fast_ramp = os.environ.get('KBE_RAMP', 'classic') == 'fast'
                batch = batch or transfer_group(n, lanes, self.video_launch_shape(state, cameras, -1, to_host=True)[1] if lanes > DEFAULT_HOST_LANES else 0, fast_ramp)
        # the staging buffers grow with |batch| (lanes * (4 + G) frames): never more frames per transfer than the video has, or than 64
        batch = int(batch)
        batch = -min(-batch, max(n, 1), 64) if batch < 0 else min(batch, max(n, 1), 64)
        need = int(self.lib.kbe_video_stage_bytes(_i(W), _i(H), _i(lanes), _i(batch)))
        if 'stage' not in state or state['stage'].numel() < need:
            state['stage'] = torch.empty(need, dtype=torch.uint8, device=dev)
        if 'copy_stream' not in state:
            state['copy_stream'] = torch.cuda.Stream(device=dev)
            state['lane_streams'] = [None] + lane_streams_of(dev, state['lanes'] - 1)
        lane_streams = (ctypes.c_void_p * MAX_LANES)(*[None if st is None else st.cuda_stream for st in state['lane_streams']])
        focals = (ctypes.c_double * max(n, 1))(*[float(c[0]) for c in cameras])
        shifts = (ctypes.c_float * max(3 * n, 1))(*[float(v) for c in cameras for v in c[1]])
        cw, ch = (0, 0) if crop is None else (int(crop[0]), int(crop[1]))
        copy_stream = ctypes.c_void_p(state['copy_stream'].cuda_stream) if overlap else _stream()
        flags, group, fused = self.video_launch_shape(state, cameras, batch, to_host=not host_out.is_cuda)
        base_flags = flags
        if fused:
            self._pack(state)
        # KBE_VIDEO_FREE_TRANSFERS: videos that fill with the tables are bound by their rendering (the link is half idle), and
        # a lane waiting for its turn on the link only idles: bench --dolly 8.1 k frames/s delivered with turns, 9.1 k without
        # (512^2 and 2048^2 frames, whose transfers fill the link to 70 %, keep the turns: 57 vs 53 k, 2.65 vs 2.03 k)
        free = os.environ.get('KBE_FREE_TRANSFERS', 'auto')
        if not host_out.is_cuda and (free == '1' or (free == 'auto' and flags & 1)):
            flags |= 8
        if os.environ.get('KBE_EVEN_GROUPS') == '1':        # (dev) transfer groups of one size instead of the ramp 1, 2, 4, ...
            flags |= 16
        if os.environ.get('KBE_RAMP', 'classic') == 'fast':    # KBE_VIDEO_FAST_RAMP: transfer groups of 1, 3, 7, 15, ... frames
            flags |= 1024
        if os.environ.get('KBE_AHEAD') == '0':              # KBE_VIDEO_NO_AHEAD: every group of the fused route keeps its own placement launch
            flags |= 512
        flags |= fused_build_bits(video=True)               # KBE_FUSED_CAP: KBE_VIDEO_FUSED_LEAN / _ROOMY
        if not host_out.is_cuda and batch < 0 and handoff_by_sdma():
            flags |= 8192                                   # KBE_VIDEO_SDMA
            if os.environ.get('KBE_INJECT_HANDOFF_FAULT') == '1':
                flags |= 32768                              # KBE_VIDEO_INJECT_FAULT (test hook: tests/test_hip_parity.py)
            if os.environ.get('KBE_INJECT_HANDOFF_TIMEOUT') == '1':
                flags |= 65536                              # KBE_VIDEO_INJECT_TIMEOUT (test hook)
        keep_flags = flags & ~base_flags                    # the switches set above, should the launch shape be taken again
        scratch = state['scratch']
        if group > 1:
            # n scratch sets per lane in use, allocated on first use -- 224 MB each at 1024^2, 0.9 GB at 2048^2 (most of it the
            # bucket / spill area): a launch shape that would take more than the budget (KBE_SCRATCH_BUDGET_MB, default half of
            # what is free, never less than one set per lane) falls back to fewer frames per launch
            # (asked only when the sets the cloud already holds do not do: the budget reads the device's free memory, ~10 us of every call)
            fits = group
            if 'scratch_groups' not in state or state['scratch_groups'].numel() < group * lanes * self.scratch_stride(state) or os.environ.get('KBE_SCRATCH_BUDGET_MB'):
                max_sets = self.scratch_set_budget(state)
                while fits > 1 and fits * lanes > max_sets:
                    fits = max(1, fits // 2)
            if fits != group:
                flags, group, fused = self.video_launch_shape(state, cameras, batch, to_host=not host_out.is_cuda, max_group=fits)
                flags |= keep_flags
        state['video_sets'] = group * lanes
        if group > 1:
            scratch, _ = self.group_scratch(state, group * lanes)
        self._check(self.lib.kbe_render_video(_ptr(state['points']), _ptr(state['image']), _ptr(state['depth']), _i(state['N']),
                                              _i(W), _i(H), _d(float(baseline)), _i(n), focals, shifts, _i(cw), _i(ch),
                                              _ptr(scratch, torch.uint8), _ptr(state['stage'], torch.uint8), _i(batch),
                                              ctypes.c_void_p(host_out.data_ptr()), _i(state['raster_w']), _i(state['raster_n']),
                                              _ptr(state['packed'], torch.uint8) if fused else None, _d(state['cloud_focal']),
                                              _i(flags), _stream(), copy_stream, _i(lanes), lane_streams, _d(self.near_depth(state) if fused else 0.0)), 'kbe_render_video')
        return host_out

    def handoff_status(self):
        """After synchronising a delivered video's stream: raises KbeError when an SDMA hand-off of this process gave up waiting for its
        engine (kbe_video_handoff_status: the frames of that video are not all in host memory)."""
        self._check(self.lib.kbe_video_handoff_status(), 'kbe_video_handoff_status')

    def generate_mask_raw(self, points, shift, W, H, focal, baseline, want_tables=False):
        """generate_mask's kernel (common.py:696-817) -> masks [B,1,N] (and zee [B,1,H,W], ids [B,H,W] int32)."""
        points = _f32c(points)
        B, _, N = points.shape
        dev = points.device
        shift = _f32c(shift).reshape(B, 3).to(dev)
        keys = torch.empty(B * H * W, dtype=torch.int64, device=dev)
        winner = torch.empty(B, max(N, 1), dtype=torch.int32, device=dev)
        masks = torch.empty(B, 1, N, dtype=torch.float32, device=dev)
        zee = torch.empty(B, 1, H, W, dtype=torch.float32, device=dev) if want_tables else None
        ids = torch.empty(B, H, W, dtype=torch.int32, device=dev) if want_tables else None
        self._check(self.lib.kbe_generate_mask(_ptr(points), _ptr(shift), _i(B), _i(N), _i(int(W)), _i(int(H)), _d(float(focal)),
                                               _d(float(baseline)), _ptr(keys, torch.int64), _ptr(winner, torch.int32), _ptr(masks),
                                               _ptr(zee), _ptr(ids, torch.int32), _stream()), 'kbe_generate_mask')
        return (masks, zee, ids) if want_tables else masks

    def generate_mask(self, points, shift, W, H, focal, baseline):
        """common.py:689-830: the ownership mask as an image (N == H*W), median-5 filtered (:829)."""
        masks = self.generate_mask_raw(points, shift, W, H, focal, baseline)
        return self.spatial_filter(masks.view(-1, 1, int(H), int(W)), 'median-5')

    def zkeys_clear(self, zkeys):
        self._check(self.lib.kbe_zkeys_clear(_ptr(zkeys, torch.int32), _z(zkeys.numel()), _stream()), 'kbe_zkeys_clear')

    def frame_u8(self, render):
        render = _f32c(render)
        _, C, H, W = render.shape
        out = torch.empty(H, W, 3, dtype=torch.uint8, device=render.device)
        self._check(self.lib.kbe_frame_u8(_ptr(render), _i(W), _i(H), _ptr(out, torch.uint8), _stream()), 'kbe_frame_u8')
        return out

    def crop_resize_u8(self, frame, crop_w, crop_h):
        H, W, _ = frame.shape
        out = torch.empty_like(frame)
        self._check(self.lib.kbe_crop_resize_u8(_ptr(frame, torch.uint8), _i(W), _i(H), _i(int(crop_w)), _i(int(crop_h)),
                                                _ptr(out, torch.uint8), _stream()), 'kbe_crop_resize_u8')
        return out

    # -- torch glue ---------------------------------------------------------------------
    def depth_to_points(self, depth, focal, valid=None):
        depth = _f32c(depth)
        B, _, H, W = depth.shape
        out = torch.empty(B, 3, H, W, dtype=torch.float32, device=depth.device)
        self._check(self.lib.kbe_depth_to_points(_ptr(depth), _ptr(None if valid is None else _f32c(valid)), _i(B), _i(W),
                                                 _i(H), _d(float(focal)), _ptr(out), _stream()), 'kbe_depth_to_points')
        return out

    def shift_points(self, points, shift3):
        points = _f32c(points)
        B, _, N = points.shape
        out = torch.empty_like(points)
        self._check(self.lib.kbe_shift_points(_ptr(points), _i(B), _i(N), _shift(shift3), _ptr(out), _stream()),
                    'kbe_shift_points')
        return out

    def spatial_filter(self, x, kind):
        code = {'laplacian': 0, 'median-3': 3, 'median-5': 5}.get(kind)
        if code is None:
            return None       # the reference returns None for unknown types (common.py:395,425)
        x = _f32c(x)
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        self._check(self.lib.kbe_spatial_filter(_ptr(x), _i(B * C), _i(W), _i(H), _i(code), _ptr(out), _stream()),
                    'kbe_spatial_filter')
        return out

    def laplacian_valid(self, disparity, scale, threshold):
        """(|laplacian(disparity / scale)| < threshold).float(); `scale` is a 0-dim device tensor."""
        x = _f32c(disparity)
        B, C, H, W = x.shape
        out = torch.empty_like(x)
        scale = _f32c(scale).reshape(1)
        self._check(self.lib.kbe_laplacian_valid(_ptr(x), _ptr(scale), _i(B * C), _i(W), _i(H), _f(float(threshold)), _ptr(out),
                                                 _stream()), 'kbe_laplacian_valid')
        return out

    def prelu_mask(self, x, slope, mask=None, out=None):
        """prelu(x, slope) * mask in one pass (kbe_prelu_mask); mask [B,1,H,W] or None."""
        x = _f32c(x)
        B, C, H, W = x.shape
        if mask is not None and tuple(mask.shape) != (B, 1, H, W):
            raise KbeError('prelu_mask: the mask must be [B,1,H,W] = %s, got %s' % ((B, 1, H, W), tuple(mask.shape)))
        out = torch.empty_like(x) if out is None else out
        self._check(self.lib.kbe_prelu_mask(_ptr(x), _ptr(_f32c(slope)), _ptr(None if mask is None else _f32c(mask)), _i(B), _i(C), _i(H), _i(W), _ptr(out),
                                            _stream()), 'kbe_prelu_mask')
        return out

    def bias_act(self, x, bias=None, slope=None, res1=None, res2=None, out=None):
        """act(x + bias[c]) + res1 + res2 in one pass (kbe_bias_act): x [B,C,H,W] contiguous fp32; every other operand optional.
        `out` may be `x` itself (a convolution's fresh output)."""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
        B, C, H, W = x.shape
        out = torch.empty_like(x) if out is None else out
        res1, res2 = (None if r is None else _f32c(r) for r in (res1, res2))
        assert all(r is None or r.shape == x.shape for r in (res1, res2))
        self._check(self.lib.kbe_bias_act(_ptr(x), _ptr(None if bias is None else _f32c(bias)), _ptr(None if slope is None else _f32c(slope)), _ptr(res1), _ptr(res2),
                                          _i(B), _i(C), _i(H), _i(W), _ptr(out), _stream()), 'kbe_bias_act')
        return out

    def upsample2x_act(self, x, slope=None):
        """prelu(bilinear x2 upsampling of x, align_corners=False) in one pass (kbe_upsample2x_act); slope [C] or None."""
        assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
        B, C, H, W = x.shape
        out = torch.empty(B, C, 2 * H, 2 * W, dtype=torch.float32, device=x.device)
        self._check(self.lib.kbe_upsample2x_act(_ptr(x), _ptr(None if slope is None else _f32c(slope)), _i(B), _i(C), _i(H), _i(W), _ptr(out), _stream()), 'kbe_upsample2x_act')
        return out

    pconv_epilogue_adds_bias = True         # (PartialConv2d.forward: run the convolution without its bias)

    def pconv_epilogue(self, raw, bias, mask, kernel_size, stride, padding, in_channels=None, in_size=None, act_slope=None, residual=None,
                       raw_without_bias=False):
        """raw [B,Cout,Ho,Wo] = conv(x * mask); mask [B,Cin|1,H,W] or None (then in_channels/in_size say what x was).
        act_slope [Cout] / residual [B,Cout,Ho,Wo] (optional): out = prelu(out + residual) in the same pass (include/kbe.h).
        raw_without_bias: the convolution was run without its bias; the epilogue adds it first (same values, one pass less)."""
        raw = _f32c(raw)
        B, Cout, Ho, Wo = raw.shape
        if mask is not None:
            mask = _f32c(mask)
            _, Cm, H, W = mask.shape
            Cin = Cm if in_channels is None else int(in_channels)
        else:
            Cm, Cin, (H, W) = 1, int(in_channels), in_size
        out = torch.empty_like(raw)
        um = torch.empty(B, 1, Ho, Wo, dtype=torch.float32, device=raw.device)
        if residual is not None:
            residual = _f32c(residual)
            assert residual.shape == raw.shape
        self._check(self.lib.kbe_pconv_epilogue(_ptr(raw), _ptr(None if bias is None else _f32c(bias)), _ptr(mask), _i(Cm), _i(B),
                                                _i(Cin), _i(H), _i(W), _i(Cout), _i(Ho), _i(Wo), _i(int(kernel_size)),
                                                _i(int(stride)), _i(int(padding)), _ptr(out), _ptr(um),
                                                _ptr(None if act_slope is None else _f32c(act_slope)), _ptr(residual),
                                                _i(1 if raw_without_bias and bias is not None else 0), _stream()),
                    'kbe_pconv_epilogue')
        return out, um


class HipSerialScheduleKernels:
    """KBE_DEGRID=serial: the frames the REFERENCE-RUN fixtures hold (tests/golden/kenburns_*.npz: the reference's kernel text
    executed one element after the other), from the HIP library.

    The reference's updateDegrid (common.py:525-568) rewrites the z-buffer in place while its neighbours read it; what a CUDA
    run produces lies somewhere between the serial index order and the out-of-place (Jacobi) schedule and is not reproducible
    from run to run.  The product's default is Jacobi -- the fused tile launch, the only deterministic
    parallel schedule: against the serial one it moves 0.1-0.65 % of a frame's pixels (32-59 dB by scene).  This kernel set is the
    other end of that range: every op is still the HIP library's, but render_pointcloud / render_frame run stage by stage --
    kbe_zsplat, kbe_degrid_serial (a wavefront sweep with exactly the serial schedule's data dependences), kbe_accumulate,
    kbe_normalize, kbe_fill_disocclusion, kbe_frame_u8 -- one frame at a time (no native video loop).  Byte for byte the frames
    of the fixtures up to the order of the fp32 atomics (one count on < 0.2 % of the values); ~2 ms per 1024^2 frame instead
    of 21 us: for comparisons with a serial execution of the reference, not for production."""
    name = 'hip-serial'

    def __init__(self, K, keep_zee=False):
        self.K, self.zee_log = K, ([] if keep_zee else None)

    def __getattr__(self, name):
        if name == 'render_video':            # forces common.render_frames onto the per-frame loop below
            raise AttributeError(name)
        return getattr(self.K, name)

    def render_pointcloud(self, points, data, W, H, focal, baseline):
        zkeys, _ = self.K.zsplat(points, W, H, focal, baseline)                                     # common.py:435-507
        zee = self.K.degrid_serial(zkeys=zkeys)                                                     # :525-568, serial schedule
        if self.zee_log is not None:
            self.zee_log.append(zee)
        return self.K.normalize(self.K.accumulate(points, data, zee, focal, baseline))              # :586-669, :686

    def prepare_cloud(self, points, image, depth, W, H, focal=None, raster=None, **kw):
        return {'points': points.reshape(1, 3, -1), 'data': torch.cat([image.reshape(1, 3, -1), depth.reshape(1, 1, -1)], 1), 'W': W, 'H': H}

    def render_frame(self, state, shift3, focal, baseline, fill_rect=None, **kw):
        pts = self.K.shift_points(state['points'], shift3)                                          # common.py:238-244
        render, existing = self.render_pointcloud(pts, state['data'], state['W'], state['H'], focal, baseline)   # :246-251
        filled = self.K.fill_disocclusion(render, render[:, 3:4] * (existing > 0.0).float())       # :253
        return self.K.frame_u8(filled)                                                              # :255


_kernels = None
_serial_kernels = None


def degrid_schedule():
    """KBE_DEGRID=jacobi (default: the fused tile launch) | serial (HipSerialScheduleKernels)."""
    v = os.environ.get('KBE_DEGRID', 'jacobi')
    if v not in ('jacobi', 'serial'):
        raise KbeError('KBE_DEGRID=%r: jacobi (default) or serial' % v)
    return v


def kernels():
    """The process-wide kernel set (HIP).  Raises KbeError when the extension is missing.  KBE_DEGRID=serial: the same library
    behind HipSerialScheduleKernels (read per call, like every other switch of the host side)."""
    global _kernels, _serial_kernels
    if _kernels is None:
        _kernels = HipKernels()
    if degrid_schedule() == 'serial':
        if _serial_kernels is None:
            _serial_kernels = HipSerialScheduleKernels(_kernels)
        return _serial_kernels
    return _kernels
