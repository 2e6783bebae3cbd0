"""Seeded synthetic RGBD inputs and deterministic weights.

The reference ships no test inputs and its checkpoints are a network download
(``download.sh``), so every test, fixture and benchmark in this repo runs on the
inputs defined here:

* RGB: uniform [0, 1).
* disparity: smooth ramp + sinusoid + 1..3 rectangles at the maximum, shifted to
  >= 0 and scaled so that max == baseline (what ``utils/pipeline.py:79-81`` does
  to the refined disparity), or a white-noise variant (degrid stress case).
* camera constants F = 512, B = 120 (``utils/pipeline.py:26-27``).

Everything is generated with numpy's PCG64 on the host so that the very same
bits are produced in the build container and on the GPU box.
"""
import math
import zlib

import numpy as np
import torch

FOCAL = 1024.0 / 2   # utils/pipeline.py:26
BASELINE = 120       # utils/pipeline.py:27 (an *int* there; see include/kbe.h)


def photo_like(image, cell=24):
    """A photograph's statistics from white noise: the seeded colours low-passed -- a coarse grid of them (one per `cell` pixels)
    interpolated bilinearly -- plus a tenth of the noise as texture.  Pure numpy on the host, float64 inside: the same bits
    in the build container and on the GPU box."""
    _, C, H, W = image.shape
    gh, gw = H // cell + 2, W // cell + 2
    coarse = image[0, :, :gh * 1, :gw * 1].astype(np.float64) if (gh <= H and gw <= W) else None
    if coarse is None:
        return image
    ys = (np.arange(H, dtype=np.float64) + 0.5) / cell
    xs = (np.arange(W, dtype=np.float64) + 0.5) / cell
    y0 = np.minimum(np.floor(ys).astype(np.int64), gh - 2)
    x0 = np.minimum(np.floor(xs).astype(np.int64), gw - 2)
    fy = (ys - y0)[None, :, None]
    fx = (xs - x0)[None, None, :]
    a = coarse[:, y0][:, :, x0]
    b = coarse[:, y0][:, :, x0 + 1]
    c = coarse[:, y0 + 1][:, :, x0]
    d = coarse[:, y0 + 1][:, :, x0 + 1]
    low = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    out = 0.9 * low + 0.1 * image[0].astype(np.float64)
    return np.ascontiguousarray(out.astype(np.float32))[None]


def make_rgbd(height, width, seed=0, kind='smooth', baseline=BASELINE, colours='noise'):
    """Returns (image[1,3,H,W], disparity[1,1,H,W]) as float32 CPU tensors.  colours: 'noise' (uniform [0, 1): SURVEY.md 8d)
    or 'photo' (the same noise low-passed: photo_like)."""
    rng = np.random.default_rng(seed)
    image = rng.random((1, 3, height, width), dtype=np.float32)
    if colours == 'photo':
        image = photo_like(image)
    elif colours != 'noise':
        raise ValueError('unknown colours: ' + str(colours))
    ys = np.arange(height, dtype=np.float64)[:, None]
    xs = np.arange(width, dtype=np.float64)[None, :]
    if kind == 'smooth':
        # 20 + 60*(y/H) + 10*sin(x/37) at 1024 wide; period scaled with the width
        disp = 20.0 + 60.0 * (ys / height) + 10.0 * np.sin(xs * (1024.0 / 37.0) / width)
        disp = np.broadcast_to(disp, (height, width)).copy()
        for _ in range(int(rng.integers(1, 4))):
            rh = int(rng.integers(max(2, height // 8), max(3, height // 3)))
            rw = int(rng.integers(max(2, width // 8), max(3, width // 3)))
            y0 = int(rng.integers(0, height - rh))
            x0 = int(rng.integers(0, width - rw))
            disp[y0:y0 + rh, x0:x0 + rw] = 120.0
    elif kind == 'noise':
        disp = rng.random((height, width)) * 100.0 + 20.0
    elif kind == 'flat':
        disp = np.full((height, width), 60.0)
    else:
        raise ValueError('unknown scene kind: ' + str(kind))
    disp = disp.astype(np.float32)
    disp = disp - min(float(disp.min()), 0.0)
    disp = disp / disp.max() * np.float32(baseline)
    return torch.from_numpy(image), torch.from_numpy(disp.astype(np.float32))[None, None]


def depthrange_of(depth, border=128):
    """(minVal, maxVal, minLoc(x, y), maxLoc(x, y)) of the border-cropped depth.

    Stand-in for ``cv2.minMaxLoc(depth[128:-128, 128:-128])`` at
    ``utils/pipeline.py:96``.  The reference's slice is empty when H or W <= 256;
    here the border shrinks to ``min(128, H // 4, W // 4)`` (documented deviation,
    SURVEY.md 8a-a4).  Ties resolve to the first element in raster order, which
    is what OpenCV's scan does.
    """
    d = depth.detach().cpu().numpy().reshape(depth.shape[-2], depth.shape[-1])
    b = min(border, d.shape[0] // 4, d.shape[1] // 4)
    crop = d[b:d.shape[0] - b, b:d.shape[1] - b] if b > 0 else d
    imin = int(np.argmin(crop))
    imax = int(np.argmax(crop))
    w = crop.shape[1]
    return (float(crop.flat[imin]), float(crop.flat[imax]),
            (imin % w, imin // w), (imax % w, imax // w))


def default_windows(height, width, dolly=False):
    """The crop windows ``kbe.py:128-140`` picks when none are given."""
    if not dolly:
        start = (width / 2.15, height / 2.15, int(math.floor(0.90 * width)), int(math.floor(0.90 * height)))
        end = (width / 1.85, height / 1.85, int(math.floor(0.85 * width)), int(math.floor(0.85 * height)))
    else:
        start = (width / 2, height / 2, int(math.floor(0.8 * width)), int(math.floor(0.8 * height)))
        end = (width / 2, height / 2, int(math.floor(0.3 * width)), int(math.floor(0.3 * height)))
    keys = ('dblCenterU', 'dblCenterV', 'intCropWidth', 'intCropHeight')
    return dict(zip(keys, start)), dict(zip(keys, end))


def seeded_fill_(module, seed=0):
    """Deterministic, construction-order-independent weights for any nn.Module.

    Every state-dict entry is filled from its own generator seeded with
    ``crc32(name) ^ seed``: conv / linear weights uniform in +-sqrt(3 / fan_in)
    (variance-preserving), 1-D ``weight`` (PReLU slopes, norm scales) = 0.25 + small
    jitter, biases uniform in +-0.05, running_var = 1.  Used wherever the reference
    loads a checkpoint that cannot be downloaded here, and by the golden
    generator so that reference and rebuilt modules carry identical weights.
    """
    state = module.state_dict()
    for name in sorted(state):
        t = state[name]
        if not t.dtype.is_floating_point:
            continue
        rng = np.random.default_rng((zlib.crc32(name.encode()) ^ seed) & 0xFFFFFFFF)
        shape = tuple(t.shape)
        if t.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            bound = math.sqrt(3.0 / fan_in)
            v = rng.uniform(-bound, bound, size=shape)
        elif name.endswith('running_var'):
            v = np.ones(shape)
        elif name.endswith('running_mean'):
            v = np.zeros(shape)
        elif name.endswith('bias'):
            v = rng.uniform(-0.05, 0.05, size=shape)
        else:
            v = 0.25 + rng.uniform(-0.05, 0.05, size=shape)
        t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)).reshape(shape))
    return module
