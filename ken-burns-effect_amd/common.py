"""Drop-in for the render half of the reference's ``utils/common.py``.

Same function names, argument meaning, dictionary keys (``objectCommon`` /
``objectSettings``) and return types as ``/root/reference/utils/common.py:16-263,
382-426, 428-686, 833-937``; the CUDA-string kernels and the CuPy launcher
(``:267-380``) are replaced by the precompiled gfx950 library behind
:mod:`ken_burns_effect_amd._native` (C ABI in ``include/kbe.h``).

What differs from the reference, on purpose:

* the frame loop of :func:`process_kenburns` does not materialise the shifted point
  cloud, the concatenated data tensor, the z-buffer fill, the normalised render or the
  filled float image per frame -- one fused call per frame reads the resident packed
  cloud and writes the uint8 frame (``kbe_render_frame``); the crop + resize of
  ``common.py:256-257`` runs on the device and frames reach the host through one
  pinned buffer with a single synchronisation at the end.
* the unused render of the set-up loop (``common.py:208-215``) is skipped.
* degrid uses the deterministic out-of-place schedule.
* ``objectDepthrange`` for images <= 256 px uses a shrunken border (see
  :func:`ken_burns_effect_amd.synthetic.depthrange_of`).

There is no CPU code path here: the kernel set comes from ``_native.kernels()`` and
raises when the HIP library is missing.  (Tests substitute ``_kernel_set`` with the
oracle explicitly to exercise the host logic without a GPU.)
"""
import contextlib
import math

import numpy as np
import torch

from . import _native

# hooks that play the role of the undefined globals the reference's process_load calls
# (common.py:23-25); Pipeline sets them, tests may too.
disparity_estimation = None
disparity_refinement = None

_kernel_set = None     # tests assign an oracle-backed kernel set here; the product never does


def _K():
    return _kernel_set if _kernel_set is not None else _native.kernels()


def on_device_of(t):
    """Context in which `t`'s GPU is the current device: the C ABI launches on the CURRENT device's stream, so work on a
    tensor of cuda:1 must run with device 1 current (a caller on a multi-GPU box need not call torch.cuda.set_device)."""
    device = t if isinstance(t, torch.device) else (torch.device(t) if isinstance(t, str) else t.device)
    return torch.cuda.device(device) if device.type == 'cuda' else contextlib.nullcontext()


# ---------------------------------------------------------------------------------------
# L2 render ops (reference: common.py:382-426, 428-686, 833-937)
# ---------------------------------------------------------------------------------------

def depth_to_points(tensorDepth, dblFocal):
    """[B,1,H,W] depth -> [B,3,H,W] camera-space points (common.py:382-392)."""
    return _K().depth_to_points(tensorDepth, dblFocal)


def spatial_filter(tensorInput, strType):
    """'laplacian' | 'median-3' | 'median-5' (common.py:394-426); None for other types."""
    return _K().spatial_filter(tensorInput, strType)


def render_pointcloud(tensorInput, tensorData, intWidth, intHeight, dblFocal, dblBaseline):
    """Forward-warps points [B,3,N] carrying data [B,C,N] into an HxW view (common.py:428-686).

    Returns (render [B,C,H,W] normalised by the accumulated bilinear weight,
    existing [B,1,H,W] = that weight)."""
    return _K().render_pointcloud(tensorInput, tensorData, int(intWidth), int(intHeight), dblFocal, dblBaseline)


def fill_disocclusion(tensorInput, tensorDepth):
    """Fills pixels whose depth <= 0 from the farther end of the shortest of 16 rays (common.py:833-937)."""
    return _K().fill_disocclusion(tensorInput, tensorDepth)


# ---------------------------------------------------------------------------------------
# camera path (reference: common.py:83-112 and the scalar head of both loops, :185-198 / :223-236)
# ---------------------------------------------------------------------------------------

def generate_mask(tensorInput, tensorShift, intWidth, intHeight, dblFocal, dblBaseline):
    """Disocclusion mask of view A seen from view B (common.py:689-830): for each point of the image raster
    ``tensorInput`` [B,3,H*W] moved by ``tensorShift`` [B,3,1], 1 where it still owns the pixel it lands on
    (nearest point, first in index order on ties), else 0; returned as an image [B,1,H,W] after the median-5 of
    :829.  The reference's kernel is racy; this is its serial-order result, computed deterministically
    (include/kbe.h, kbe_generate_mask)."""
    return _K().generate_mask(tensorInput, tensorShift, intWidth, intHeight, dblFocal, dblBaseline)


def _shift_vector(objectSettings, objectCommon, dblFocal):
    """The three python doubles of common.py:88-100, expression for expression."""
    depthrange = objectCommon['objectDepthrange']
    dblClosestDepth = depthrange[0] + (objectSettings['dblDepthTo'] - objectSettings['dblDepthFrom'])
    fromU, fromV = depthrange[2][0], depthrange[2][1]
    toU, toV = fromU + objectSettings['dblShiftU'], fromV + objectSettings['dblShiftV']
    halfW, halfH = objectCommon['intWidth'] / 2.0, objectCommon['intHeight'] / 2.0
    fromX = ((fromU - halfW) * dblClosestDepth) / dblFocal
    fromY = ((fromV - halfH) * dblClosestDepth) / dblFocal
    toX = ((toU - halfW) * dblClosestDepth) / dblFocal
    toY = ((toV - halfH) * dblClosestDepth) / dblFocal
    return fromX - toX, fromY - toY, objectSettings['dblDepthTo'] - objectSettings['dblDepthFrom']


def process_shift(objectSettings, objectCommon, dblFocal=None):
    """Camera translation for a crop-window shift (common.py:83-112).

    objectSettings: tensorPoints [1,3,N], dblShiftU/V, dblDepthFrom/To.
    Returns (shifted points [1,3,N], tensorShift [1,3,1])."""
    if dblFocal is None:
        dblFocal = objectCommon['dblFocal']
    points = objectSettings['tensorPoints']
    vec = _shift_vector(objectSettings, objectCommon, dblFocal)
    tensorShift = torch.tensor(vec, dtype=torch.float64).to(torch.float32).view(1, 3, 1).to(points.device)
    return _K().shift_points(points, tensorShift), tensorShift


def _camera_at(dblStep, objectSettings, objectCommon):
    """Per-step scalars shared by the set-up loop and the frame loop (common.py:182-198 == :223-236)."""
    dblFrom = 1.0 - dblStep
    dblTo = 1.0 - dblFrom
    oFrom, oTo = objectSettings['objectFrom'], objectSettings['objectTo']
    if objectSettings['dolly']:
        focalScaling = oTo['intCropWidth'] / oFrom['intCropWidth']
        focal = objectCommon['dblFocal'] * (1 - dblStep) + dblStep * objectCommon['dblFocal'] * focalScaling
    else:
        focal = objectCommon['dblFocal']
    shiftU = ((dblFrom * oFrom['dblCenterU']) + (dblTo * oTo['dblCenterU'])) - (objectCommon['intWidth'] / 2.0)
    shiftV = ((dblFrom * oFrom['dblCenterV']) + (dblTo * oTo['dblCenterV'])) - (objectCommon['intHeight'] / 2.0)
    cropW = (dblFrom * oFrom['intCropWidth']) + (dblTo * oTo['intCropWidth'])
    depthFrom = objectCommon['objectDepthrange'][0]
    depthTo = objectCommon['objectDepthrange'][0] * (cropW / max(oFrom['intCropWidth'], oTo['intCropWidth']))
    return focal, {'dblShiftU': shiftU, 'dblShiftV': shiftV, 'dblDepthFrom': depthFrom, 'dblDepthTo': depthTo}


def _shift_f32(vec):
    """python doubles -> the fp32 values torch.FloatTensor([...]) would hold (common.py:102)."""
    return [float(v) for v in np.asarray(vec, dtype=np.float64).astype(np.float32)]


# ---------------------------------------------------------------------------------------
# L3 render loop
# ---------------------------------------------------------------------------------------

def process_load(numpyImage, objectSettings, objectCommon):
    """Populates ``objectCommon`` from an HxWx3 uint8 image (common.py:16-45).

    The reference body calls two functions that do not exist in its module
    (``disparity_estimation`` / ``disparity_refinement``, :23-25); here they are the
    module-level hooks of the same names, or ``objectSettings['tensorDisparity']``
    ([1,1,H,W]) bypasses them (depth estimation bypassed, BASELINE.json configs[0]).
    Constants as in the reference: F = 1024/2, B = 40.
    """
    objectCommon['dblFocal'] = 1024 / 2.0
    objectCommon['dblBaseline'] = 40.0
    objectCommon['intWidth'] = numpyImage.shape[1]
    objectCommon['intHeight'] = numpyImage.shape[0]
    device = objectSettings.get('device', 'cuda:0') if isinstance(objectSettings, dict) else 'cuda:0'
    with on_device_of(device):
        return _process_load(numpyImage, objectSettings, objectCommon, device)


def _process_load(numpyImage, objectSettings, objectCommon, device):
    from . import synthetic
    K = _K()
    # the division runs on the host (IEEE): torch's GPU kernels multiply by the rounded reciprocal of a scalar divisor,
    # which is 1 ulp off for about half of the 256 values -- the same image must give the same cloud on every device
    tensorImage = (torch.from_numpy(np.ascontiguousarray(numpyImage.transpose(2, 0, 1))).float().unsqueeze(0) / 255.0).to(device)
    if isinstance(objectSettings, dict) and objectSettings.get('tensorDisparity') is not None:
        tensorDisparity = objectSettings['tensorDisparity'].to(device).float()
    else:
        if disparity_estimation is None or disparity_refinement is None:
            raise RuntimeError('process_load needs common.disparity_estimation / disparity_refinement hooks '
                               "or objectSettings['tensorDisparity']")
        tensorDisparity = disparity_refinement(tensorImage, disparity_estimation(tensorImage))
    tensorDisparity = tensorDisparity / tensorDisparity.max() * objectCommon['dblBaseline']
    tensorDepth = (objectCommon['dblFocal'] * objectCommon['dblBaseline']) / (tensorDisparity + 0.0000001)
    tensorValid = K.laplacian_valid(tensorDisparity, tensorDisparity.max(), 0.03)
    tensorPoints = K.depth_to_points(tensorDepth * tensorValid, objectCommon['dblFocal'])
    tensorUnaltered = K.depth_to_points(tensorDepth, objectCommon['dblFocal'])

    objectCommon['dblDispmin'] = tensorDisparity.min().item()
    objectCommon['dblDispmax'] = tensorDisparity.max().item()
    objectCommon['objectDepthrange'] = synthetic.depthrange_of(tensorDepth)
    objectCommon['tensorRawImage'] = tensorImage
    objectCommon['tensorRawDisparity'] = tensorDisparity
    objectCommon['tensorRawDepth'] = tensorDepth
    objectCommon['tensorRawPoints'] = tensorPoints.view(1, 3, -1)
    objectCommon['tensorRawUnaltered'] = tensorUnaltered.view(1, 3, -1)
    _reset_inpa(objectCommon)


def _reset_inpa(objectCommon):
    """common.py:41-44 == :176-179"""
    objectCommon['tensorInpaImage'] = objectCommon['tensorRawImage'].reshape(1, 3, -1)
    objectCommon['tensorInpaDisparity'] = objectCommon['tensorRawDisparity'].reshape(1, 1, -1)
    objectCommon['tensorInpaDepth'] = objectCommon['tensorRawDepth'].reshape(1, 1, -1)
    objectCommon['tensorInpaPoints'] = objectCommon['tensorRawPoints'].reshape(1, 3, -1)


def process_inpaint(tensorShift, objectCommon, moduleInpaint, dblFocal):
    """Inpaints the view displaced by ``tensorShift`` and appends the hole pixels as new points
    (common.py:47-81, single-module branch :63-80; the list branch :50-62 is dead in the
    reference -- undefined name at :69 -- and raises here too)."""
    if isinstance(moduleInpaint, (list, tuple)):
        raise NotImplementedError('two-network inpainting is broken in the reference (common.py:50-69) and not provided')
    K = _K()
    objectInpainted = moduleInpaint.pointcloud_inpainting(objectCommon['tensorRawImage'], objectCommon['tensorRawDisparity'],
                                                          tensorShift, objectCommon, dblFocal)
    disparity = objectInpainted['tensorDisparity']
    depth = (dblFocal * objectCommon['dblBaseline']) / (disparity + 0.0000001)
    valid = K.laplacian_valid(disparity, disparity.max(), 0.03)
    points = K.depth_to_points(depth, dblFocal, valid=valid).view(1, 3, -1) - tensorShift
    objectInpainted['tensorDepth'] = depth
    objectInpainted['tensorValid'] = valid
    objectInpainted['tensorPoints'] = points

    holes = (objectInpainted['tensorExisting'] == 0.0).view(-1)     # pixels the displaced view could not see
    idx = torch.nonzero(holes, as_tuple=False).view(-1)

    def take(t, c):
        return t.reshape(1, c, -1).index_select(2, idx)

    objectCommon['tensorInpaImage'] = torch.cat([objectCommon['tensorInpaImage'], take(objectInpainted['tensorImage'], 3)], 2)
    objectCommon['tensorInpaDisparity'] = torch.cat([objectCommon['tensorInpaDisparity'], take(disparity, 1)], 2)
    objectCommon['tensorInpaDepth'] = torch.cat([objectCommon['tensorInpaDepth'], take(depth, 1)], 2)
    objectCommon['tensorInpaPoints'] = torch.cat([objectCommon['tensorInpaPoints'], take(points, 3)], 2)


def process_autozoom(objectSettings, objectCommon):
    """Picks the crop-window shift that keeps most pixels covered (common.py:114-170).
    (The reference forgets to pass objectCommon to process_shift at :146-152 and would raise;
    this version passes it.)"""
    shifts = np.linspace(-objectSettings['dblShift'], objectSettings['dblShift'], 16)
    oFrom = objectSettings['objectFrom']
    cropW = oFrom['intCropWidth'] / objectSettings['dblZoom']
    cropH = oFrom['intCropHeight'] / objectSettings['dblZoom']
    depthFrom = objectCommon['objectDepthrange'][0]
    depthTo = objectCommon['objectDepthrange'][0] * (cropW / oFrom['intCropWidth'])
    best, bestU, bestV = 0.0, None, None
    for intU in range(16):
        for intV in range(16):
            shiftU, shiftV = shifts[intV].item(), shifts[intU].item()
            if not (cropW / 2.0 <= oFrom['dblCenterU'] + shiftU <= objectCommon['intWidth'] - (cropW / 2.0)):
                continue
            if not (cropH / 2.0 <= oFrom['dblCenterV'] + shiftV <= objectCommon['intHeight'] - (cropH / 2.0)):
                continue
            points = process_shift({'tensorPoints': objectCommon['tensorRawPoints'], 'dblShiftU': shiftU, 'dblShiftV': shiftV,
                                    'dblDepthFrom': depthFrom, 'dblDepthTo': depthTo}, objectCommon)[0]
            _, existing = render_pointcloud(points, objectCommon['tensorRawImage'].reshape(1, 3, -1), objectCommon['intWidth'],
                                            objectCommon['intHeight'], objectCommon['dblFocal'], objectCommon['dblBaseline'])
            covered = (existing > 0.0).float().sum().item()
            if best < covered:
                best, bestU, bestV = covered, shiftU, shiftV
    return {'dblCenterU': oFrom['dblCenterU'] + bestU, 'dblCenterV': oFrom['dblCenterV'] + bestV,
            'intCropWidth': int(round(oFrom['intCropWidth'] / objectSettings['dblZoom'])),
            'intCropHeight': int(round(oFrom['intCropHeight'] / objectSettings['dblZoom']))}


def build_pointcloud(objectSettings, objectCommon, moduleInpaint):
    """The set-up loop of process_kenburns (common.py:175-220): two end poses, each inpainted
    and appended.  Serial by nature (pass 2 sees the points pass 1 appended)."""
    _reset_inpa(objectCommon)
    # (the two passes hand pointcloud_inpainting the same image and disparity: what depends on those alone is kept between them and
    # released on the way out, whatever happens in between)
    keeping = moduleInpaint.keeping_source() if hasattr(moduleInpaint, 'keeping_source') else contextlib.nullcontext()
    with keeping:
        for dblStep in (0.0, 1.0):
            focal, pose = _camera_at(dblStep, objectSettings, objectCommon)
            pose['tensorPoints'] = objectCommon['tensorInpaPoints']
            vec = _shift_vector(pose, objectCommon, focal)
            tensorShift = torch.tensor(vec, dtype=torch.float64).to(torch.float32).view(1, 3, 1)
            tensorShift = tensorShift.to(objectCommon['tensorInpaPoints'].device)
            # common.py:208-215 renders this pose and discards the result; skipped.
            if not objectSettings['dolly']:
                process_inpaint(1.1 * tensorShift, objectCommon, moduleInpaint, focal)


def frame_cameras(objectSettings, objectCommon):
    """[(focal, shift3 as fp32-exact python floats)] for every step of the frame loop (common.py:222-244)."""
    cams = []
    for dblStep in objectSettings['dblSteps']:
        focal, pose = _camera_at(dblStep, objectSettings, objectCommon)
        cams.append((focal, _shift_f32(_shift_vector(pose, objectCommon, focal))))
    return cams


def crop_size(objectSettings):
    oFrom, oTo = objectSettings['objectFrom'], objectSettings['objectTo']
    return (max(oFrom['intCropWidth'], oTo['intCropWidth']), max(oFrom['intCropHeight'], oTo['intCropHeight']))


def crop_window(W, H, crop_w, crop_h):
    """Inclusive pixel rectangle the centred crop of common.py:256 reads (cv2.getRectSubPix samples
    pixels floor(c) .. floor(c) + size, c = W/2 - (size-1)/2), padded by one pixel."""
    x0 = int(math.floor(W / 2.0 - (crop_w - 1) * 0.5)) - 1
    y0 = int(math.floor(H / 2.0 - (crop_h - 1) * 0.5)) - 1
    return (max(x0, 0), max(y0, 0), min(x0 + crop_w + 2, W - 1), min(y0 + crop_h + 2, H - 1))


def _prepared_cloud(K, objectCommon):
    """The chunked resident form of tensorInpa* (built once per cloud, reused by every frame and
    by repeated calls while the three tensors stay the same objects)."""
    tensors = tuple(objectCommon[k] for k in ('tensorInpaPoints', 'tensorInpaImage', 'tensorInpaDepth'))
    key = (K.name, objectCommon['intWidth'], objectCommon['intHeight']) + tuple((t.data_ptr(), tuple(t.shape)) for t in tensors)
    cached = objectCommon.get('_kbePreparedCloud')
    if cached is None or cached[0] != key:
        kw = {}
        if getattr(K, 'name', '') == 'hip' and objectCommon.get('objectDepthrange') is not None:
            kw['near_depth'] = objectCommon['objectDepthrange'][0]      # the closest depth process_shift works with (common.py:88)
        state = K.prepare_cloud(tensors[0], tensors[1], tensors[2], objectCommon['intWidth'], objectCommon['intHeight'],
                                objectCommon['dblFocal'], raster=objectCommon.get('_kbeCloudRaster'), **kw)
        cached = (key, state, tensors)        # keeps the tensors alive so that data_ptr stays a valid identity
        objectCommon['_kbePreparedCloud'] = cached
    if '_kbeDeliveryLanes' in objectCommon:   # measured by rank 0 and broadcast with the cloud (sharding.py): no probe on this rank
        cached[1]['delivery_lanes_hint'] = dict(objectCommon['_kbeDeliveryLanes'])
    return cached[1]


def render_frames(cameras, objectCommon, crop=None, keep_on_device=False, host_out=None, overlap=True, batch=None):
    """The frame loop proper (common.py:238-257) for a list of (focal, shift3) cameras.

    One fused kernel sequence per frame on the resident packed cloud; frames land in one
    pinned host buffer and the host synchronises once.  ``crop`` = (w, h) applies the
    device-side equivalent of cv2.getRectSubPix + cv2.resize (common.py:256-257); None
    returns the un-cropped frames.  Returns uint8 [n,H,W,3] (numpy, or a device tensor when
    ``keep_on_device``).  ``host_out``: optional pre-allocated uint8 [n,H,W,3] tensor to land the frames in
    (pinned host memory, or device memory with ``keep_on_device``; otherwise a fresh one is allocated per call)."""
    K = _K()
    W, H = objectCommon['intWidth'], objectCommon['intHeight']
    state = _prepared_cloud(K, objectCommon)
    device = objectCommon['tensorInpaPoints'].device
    n = len(cameras)
    if hasattr(K, 'render_video'):
        # the native loop: kernels + async copies enqueued from C, copies overlapped on a second stream
        # (keep_on_device: the frames stay in HBM, e.g. for a device-side encoder or the gather of the sharded path)
        if keep_on_device:
            out = torch.empty(n, H, W, 3, dtype=torch.uint8, device=device) if host_out is None else host_out
            return K.render_video(state, cameras, objectCommon['dblBaseline'], crop, host_out=out)[:n]
        out = K.render_video(state, cameras, objectCommon['dblBaseline'], crop, host_out=host_out, overlap=overlap, batch=batch)
        torch.cuda.current_stream().synchronize()
        if hasattr(K, 'handoff_status'):
            K.handoff_status()          # a hand-off that gave up on its engine is an error here, not a video with stale frames in it
        return out.numpy()
    out = torch.empty(n, H, W, 3, dtype=torch.uint8, device=device)
    rect = None if crop is None else crop_window(W, H, crop[0], crop[1])
    for i, (focal, shift3) in enumerate(cameras):
        frame = K.render_frame(state, shift3, focal, objectCommon['dblBaseline'], fill_rect=rect)
        if crop is not None:
            frame = K.crop_resize_u8(frame, crop[0], crop[1])
        out[i].copy_(frame)
    return out if keep_on_device else out.cpu().numpy()


def process_kenburns(objectSettings, objectCommon, moduleInpaint):
    """Renders the camera path ``dblSteps`` between two crop windows (common.py:172-263).

    objectSettings: dblSteps, objectFrom/objectTo {dblCenterU, dblCenterV, intCropWidth,
    intCropHeight}, boolInpaint (default True), dolly.  Returns a list of uint8 HxWx3 frames.
    Optional key ``boolCrop`` (default True): apply the crop + resize of :256-257."""
    with on_device_of(objectCommon['tensorRawPoints']):
        if 'boolInpaint' not in objectSettings or objectSettings['boolInpaint'] == True:   # noqa: E712
            build_pointcloud(objectSettings, objectCommon, moduleInpaint)
        crop = crop_size(objectSettings) if objectSettings.get('boolCrop', True) else None
        frames = render_frames(frame_cameras(objectSettings, objectCommon), objectCommon, crop)
    return [frames[i] for i in range(frames.shape[0])]
