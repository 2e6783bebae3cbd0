#!/usr/bin/env python3
"""k_crop_resize_u8 alone on a stream (common.py:256-257 on the device), us per frame: launches back to back, HIP events.  CROPS = the
crop widths to time (odd: the patch starts on a whole pixel; even: on a half pixel).  KBE_LIB_PATH: a variant build (dev aid)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ken_burns_effect_amd import _native  # noqa: E402

K = _native.kernels()
size = int(os.environ.get('SIZE', '1024'))
reps = int(os.environ.get('REPS', '200'))
rng = np.random.default_rng(3)
frame = torch.from_numpy(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).cuda()
for crop in [int(v) for v in os.environ.get('CROPS', '%d,%d' % (size * 921 // 1024, size * 921 // 1024 - 1)).split(',')]:
    out = K.crop_resize_u8(frame, crop, crop)
    want = out.clone()
    # (the launch takes less than a call through the Python binding: the C entry point with its arguments prepared)
    args = (_native._ptr(frame, torch.uint8), _native._i(size), _native._i(size), _native._i(crop), _native._i(crop), _native._ptr(out, torch.uint8), _native._stream())
    fn = K.lib.kbe_crop_resize_u8
    ts = []
    for _ in range(5):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(*args)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / reps)
    print('%d^2 frame, crop %d (%s pixel): %.2f us per frame (rounds %s), checksum %d' % (size, crop, 'whole' if (size - crop) % 2 else 'half', float(np.median(ts)), ' '.join('%.2f' % t for t in sorted(ts)), int(out.to(torch.int64).sum())), 'same as the first call' if torch.equal(out, want) else 'DIFFERS', flush=True)
