"""ctypes front-end of the CPU oracle (``oracle/kbe_oracle.c``).

TEST INFRASTRUCTURE ONLY -- imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the product package.

The functions mirror the signatures of the reference's ``utils/common.py`` so that a test
can read like the reference's own code; every tensor is a contiguous fp32 CPU tensor.
:class:`OracleKernels` exposes the same tensor-level kernel interface as the product's
``ken_burns_effect_amd._native.HipKernels`` so that tests can run the host logic
(``process_kenburns`` etc.) without a GPU by injecting it explicitly.
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libkbe_oracle.so')
_lib = None


def build(force=False):
    src = os.path.join(_HERE, 'kbe_oracle.c')
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE] + (['-B'] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        for name in ('kbo_zsplat', 'kbo_fill_zee', 'kbo_degrid_serial', 'kbo_degrid_jacobi', 'kbo_accumulate',
                     'kbo_normalize', 'kbo_fill_disocclusion', 'kbo_depth_to_points', 'kbo_shift_points',
                     'kbo_laplacian', 'kbo_median', 'kbo_frame_u8', 'kbo_pconv_epilogue', 'kbo_generate_mask'):
            getattr(_lib, name).restype = None
    return _lib


def _p(t):
    if t is None:
        return ctypes.c_void_p(0)
    assert t.device.type == 'cpu' and t.is_contiguous(), 'oracle wants contiguous CPU tensors'
    return ctypes.c_void_p(t.data_ptr())


def _f32(t):
    return t.detach().to(torch.float32).contiguous()


_i, _d, _z = ctypes.c_int, ctypes.c_double, ctypes.c_size_t


# ---------------------------------------------------------------------------------------
# stage-level entry points
# ---------------------------------------------------------------------------------------

def zsplat(points, W, H, focal, baseline, use_fma=True, want_winner=False):
    """updateZee: returns (zee[B,1,H,W], winner[B,N] int32 or None)."""
    points = _f32(points)
    B, _, N = points.shape
    zee = torch.empty(B, 1, H, W)
    lib().kbo_fill_zee(_p(zee), _z(zee.numel()))
    winner = torch.empty(B, N, dtype=torch.int32) if want_winner else None
    lib().kbo_zsplat(_p(points), _i(B), _i(N), _i(W), _i(H), _d(float(focal)), _d(float(baseline)),
                     _i(int(use_fma)), _p(zee), _p(winner))
    return zee, winner


def generate_mask_raw(points, shift, W, H, focal, baseline, use_fma=True):
    """The kernel of generate_mask (common.py:696-817) on points + shift, serial point order.
    Returns (masks[B,1,N], zee[B,1,H,W], ids[B,H,W] int32 with the bits of -1.0f where no point landed)."""
    pts = (_f32(points) + _f32(shift).reshape(points.shape[0], 3, 1)).contiguous()      # common.py:690
    B, _, N = pts.shape
    zee = torch.empty(B, 1, H, W)
    lib().kbo_fill_zee(_p(zee), _z(zee.numel()))
    ids = torch.full((B, H, W), -1.0).view(torch.int32).contiguous()                   # :694: a float tensor used as int memory
    masks = torch.zeros(B, 1, N)
    lib().kbo_generate_mask(_p(pts), _i(B), _i(N), _i(W), _i(H), _d(float(focal)), _d(float(baseline)), _i(int(use_fma)),
                            _p(zee), _p(ids), _p(masks))
    return masks, zee, ids


def generate_mask(points, shift, W, H, focal, baseline, use_fma=True):
    """common.py:689-830: the ownership mask viewed as an image (N == H*W) and median-5 filtered (:829)."""
    masks, _, _ = generate_mask_raw(points, shift, W, H, focal, baseline, use_fma)
    return spatial_filter(masks.view(-1, 1, H, W), 'median-5')


def degrid(zee, schedule='jacobi'):
    B, _, H, W = zee.shape
    if schedule == 'serial':
        out = zee.clone()
        lib().kbo_degrid_serial(_p(out), _i(B), _i(W), _i(H))
    elif schedule == 'jacobi':
        out = torch.empty_like(zee)
        lib().kbo_degrid_jacobi(_p(_f32(zee)), _p(out), _i(B), _i(W), _i(H))
    else:
        raise ValueError(schedule)
    return out


def accumulate(points, data, zee, focal, baseline, use_fma=True):
    points, data = _f32(points), _f32(data)
    B, C, N = data.shape
    _, _, H, W = zee.shape
    acc = torch.zeros(B, C + 1, H, W)
    lib().kbo_accumulate(_p(points), _p(data), _i(B), _i(N), _i(C), _p(_f32(zee)), _i(W), _i(H),
                         _d(float(focal)), _d(float(baseline)), _i(int(use_fma)), _p(acc))
    return acc


def normalize(acc):
    B, C1, H, W = acc.shape
    render = torch.empty(B, C1 - 1, H, W)
    existing = torch.empty(B, 1, H, W)
    lib().kbo_normalize(_p(_f32(acc)), _i(B), _i(C1 - 1), _i(W), _i(H), _p(render), _p(existing))
    return render, existing


def render_pointcloud(tensorInput, tensorData, intWidth, intHeight, dblFocal, dblBaseline,
                      schedule='jacobi', use_fma=True):
    """utils/common.py:428-686 as a whole."""
    zee, _ = zsplat(tensorInput, intWidth, intHeight, dblFocal, dblBaseline, use_fma)
    zee = degrid(zee, schedule)
    acc = accumulate(tensorInput, tensorData, zee, dblFocal, dblBaseline, use_fma)
    return normalize(acc)


def fill_disocclusion(tensorInput, tensorDepth):
    """utils/common.py:833-937."""
    x, d = _f32(tensorInput), _f32(tensorDepth)
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    lib().kbo_fill_disocclusion(_p(x), _p(d), _i(B), _i(C), _i(W), _i(H), _p(out))
    return out


def depth_to_points(tensorDepth, dblFocal):
    d = _f32(tensorDepth)
    B, _, H, W = d.shape
    out = torch.empty(B, 3, H, W)
    lib().kbo_depth_to_points(_p(d), _i(B), _i(W), _i(H), _d(float(dblFocal)), _p(out))
    return out


def shift_points(points, shift3):
    p = _f32(points)
    B, _, N = p.shape
    s = _f32(shift3).reshape(3)
    out = torch.empty_like(p)
    lib().kbo_shift_points(_p(p), _i(B), _i(N), _p(s), _p(out))
    return out


def spatial_filter(tensorInput, strType):
    x = _f32(tensorInput)
    B, C, H, W = x.shape
    out = torch.empty_like(x)
    if strType == 'laplacian':
        lib().kbo_laplacian(_p(x), _i(B * C), _i(W), _i(H), _p(out))
    elif strType in ('median-3', 'median-5'):
        lib().kbo_median(_p(x), _i(B * C), _i(W), _i(H), _i(int(strType[-1])), _p(out))
    else:
        return None
    return out


def frame_u8(render):
    """render [C>=3,H,W] (one sample) -> uint8 [H,W,3] as common.py:255 does on the host."""
    x = _f32(render)
    _, H, W = x.shape
    out = torch.empty(H, W, 3, dtype=torch.uint8)
    lib().kbo_frame_u8(_p(x), _i(W), _i(H), _p(out))
    return out.numpy()


def crop_resize_u8(frame, crop_w, crop_h):
    """cv2.getRectSubPix(frame, (crop_w, crop_h), (W/2, H/2)) followed by cv2.resize(.., (W, H), INTER_LINEAR)
    (common.py:256-257) restated from the OpenCV 8-bit algorithms: 16-bit fixed-point
    sub-pixel weights, then 11-bit fixed-point bilinear coefficients with src = (dst + 0.5) * scale - 0.5.
    PARITY UNPINNED: OpenCV is not available in this image; this only checks the HIP kernel against the
    same written-down algorithm."""
    H, W, _ = frame.shape
    img = frame.astype(np.int64)
    cx = np.float32(W) / np.float32(2.0) - np.float32(crop_w - 1) * np.float32(0.5)
    cy = np.float32(H) / np.float32(2.0) - np.float32(crop_h - 1) * np.float32(0.5)
    ipx, ipy = int(np.floor(cx)), int(np.floor(cy))
    a, b = np.float32(cx - np.float32(ipx)), np.float32(cy - np.float32(ipy))
    one = np.float32(1.0)
    s16 = np.float32(65536.0)
    a11 = int(np.rint((one - a) * (one - b) * s16)); a12 = int(np.rint(a * (one - b) * s16))
    a21 = int(np.rint((one - a) * b * s16)); a22 = int(np.rint(a * b * s16))
    xs0 = np.clip(ipx + np.arange(crop_w), 0, W - 1); xs1 = np.clip(ipx + np.arange(crop_w) + 1, 0, W - 1)
    ys0 = np.clip(ipy + np.arange(crop_h), 0, H - 1); ys1 = np.clip(ipy + np.arange(crop_h) + 1, 0, H - 1)
    patch = (img[ys0][:, xs0] * a11 + img[ys0][:, xs1] * a12 + img[ys1][:, xs0] * a21 + img[ys1][:, xs1] * a22 + (1 << 15)) >> 16

    def coeffs(dst_n, src_n, horizontal):
        d = np.arange(dst_n, dtype=np.float64)
        f = ((d + 0.5) * (float(src_n) / dst_n) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if horizontal:
            # resize.cpp clamps the COLUMN taps and zeroes their fraction (xmin / xmax) ...
            lo = s < 0
            f[lo] = 0; s[lo] = 0
            hi = s >= src_n - 1
            f[hi] = 0; s[hi] = src_n - 1
        # ... but keeps the fraction of the ROW taps and only clips the two row indices: above the first / below the last
        # row both taps read the same row with weights (1 - f, f), which rounds differently from (1, 0) by up to one count
        # (found by the second restatement, tests/opencv_8u_restatement.c)
        c0 = np.rint((one - f) * np.float32(2048.0)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048.0)).astype(np.int64)
        return np.clip(s, 0, src_n - 1), np.clip(s + 1, 0, src_n - 1), c0, c1

    sx, sx1, ax0, ax1 = coeffs(W, crop_w, True)
    sy, sy1, by0, by1 = coeffs(H, crop_h, False)
    rows = patch[:, sx] * ax0[None, :, None] + patch[:, sx1] * ax1[None, :, None]          # [crop_h, W, 3], x2048
    r0, r1 = rows[sy], rows[sy1]
    v = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def pconv_epilogue(raw, bias, mask, kernel_size, stride, padding, in_channels=None, in_size=None):
    raw = _f32(raw)
    B, Cout, Ho, Wo = raw.shape
    if mask is not None:
        mask = _f32(mask)
        _, Cm, H, W = mask.shape
        Cin = Cm if in_channels is None else int(in_channels)
    else:
        Cm, Cin, (H, W) = 1, int(in_channels), in_size
    out = torch.empty_like(raw)
    um = torch.empty(B, 1, Ho, Wo)
    lib().kbo_pconv_epilogue(_p(raw), _p(None if bias is None else _f32(bias)), _p(mask), _i(Cm), _i(B), _i(Cin), _i(H), _i(W),
                             _i(Cout), _i(Ho), _i(Wo), _i(kernel_size), _i(stride), _i(padding), _p(out), _p(um))
    return out, um


# ---------------------------------------------------------------------------------------
# kernel-set object for host-logic tests (same interface as _native.HipKernels)
# ---------------------------------------------------------------------------------------

class OracleKernels:
    """Tensor-level kernel set backed by the oracle; injected by tests, never by the product."""

    name = 'oracle'

    def __init__(self, schedule='jacobi', use_fma=True):
        self.schedule, self.use_fma = schedule, use_fma

    def render_pointcloud(self, points, data, W, H, focal, baseline):
        return render_pointcloud(points, data, W, H, focal, baseline, self.schedule, self.use_fma)

    def fill_disocclusion(self, x, depth):
        return fill_disocclusion(x, depth)

    def depth_to_points(self, depth, focal, valid=None):
        return depth_to_points(depth if valid is None else _f32(depth) * _f32(valid), focal)

    def shift_points(self, points, shift3):
        return shift_points(points, shift3)

    def spatial_filter(self, x, kind):
        return spatial_filter(x, kind)

    def generate_mask(self, points, shift, W, H, focal, baseline):
        return generate_mask(points, shift, W, H, focal, baseline, self.use_fma)

    def laplacian_valid(self, disparity, scale, threshold):
        lap = spatial_filter(_f32(disparity) / scale, 'laplacian')
        return (lap.abs() < threshold).float()

    def frame_u8(self, render):
        return torch.from_numpy(frame_u8(render[0]))

    def prepare_cloud(self, points, image, depth, W, H, focal=None, raster=None):
        return {'points': _f32(points).reshape(1, 3, -1), 'image': _f32(image).reshape(1, 3, -1),
                'depth': _f32(depth).reshape(1, 1, -1), 'W': int(W), 'H': int(H)}

    def render_frame(self, state, shift3, focal, baseline, want_float=False, fill_rect=None):
        """shift -> render(4 ch) -> fill -> uint8, the per-frame body of common.py:238-255."""
        pts = shift_points(state['points'], torch.tensor(shift3, dtype=torch.float32))
        data = torch.cat([state['image'], state['depth']], 1)
        render, existing = self.render_pointcloud(pts, data, state['W'], state['H'], focal, baseline)
        filled = fill_disocclusion(render, render[:, 3:4] * (existing > 0.0).float())
        frame = torch.from_numpy(frame_u8(filled[0]))
        return (frame, filled, existing) if want_float else frame

    def crop_resize_u8(self, frame, crop_w, crop_h):
        return torch.from_numpy(crop_resize_u8(frame.numpy(), crop_w, crop_h))

    def pconv_epilogue(self, raw, bias, mask, kernel_size, stride, padding, in_channels=None, in_size=None):
        return pconv_epilogue(raw, bias, mask, kernel_size, stride, padding, in_channels, in_size)
