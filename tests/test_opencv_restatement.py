"""Cross-check of the OpenCV 8-bit paths of common.py:256-257 (cv2.getRectSubPix + cv2.resize INTER_LINEAR): the oracle's
numpy restatement (what the HIP kernel k_crop_resize_u8 is tested against, tests/test_hip_parity.py) versus a second,
independently written restatement that follows the structure of OpenCV's own implementation
(tests/opencv_8u_restatement.c).  OpenCV is not installed here: agreement of two restatements catches transcription
errors, it does not pin parity with cv2 -- DESIGN.md keeps that row "parity unpinned".  A third party checks the GEOMETRY (Pillow's
bilinear resize: within one count on noise where both interpolate once).  The last two tests pin the arithmetic the day an
image with OpenCV runs them: they call cv2 itself, exactly as common.py:256-257 and pipeline.py:96 do, and are SKIPPED without it."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def cvr(tmp_path_factory):
    so = str(tmp_path_factory.mktemp('cvr') / 'libcvr.so')
    subprocess.check_call(['gcc', '-O2', '-std=c99', '-fPIC', '-shared', '-ffp-contract=off', os.path.join(HERE, 'opencv_8u_restatement.c'), '-o', so, '-lm'])
    return ctypes.CDLL(so)


def _second(cvr, frame, cw, ch):
    H, W, _ = frame.shape
    out = np.empty_like(frame)
    cvr.cvr_crop_resize_u8(frame.ctypes.data_as(ctypes.c_void_p), W, H, cw, ch, out.ctypes.data_as(ctypes.c_void_p))
    return out


@pytest.mark.parametrize('H,W,cw,ch', [(64, 96, 86, 57), (64, 96, 85, 58), (1024, 1024, 921, 921), (512, 512, 460, 460), (40, 56, 50, 36),
                                       (37, 53, 47, 33), (48, 64, 64, 48), (33, 47, 1, 1), (120, 90, 89, 119), (256, 320, 288, 230)])
def test_two_restatements_of_the_opencv_paths_agree(oracle, cvr, H, W, cw, ch):
    rng = np.random.default_rng(H * 1000 + W)
    frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    a = oracle.crop_resize_u8(frame, cw, ch)
    b = _second(cvr, np.ascontiguousarray(frame), cw, ch)
    assert a.shape == b.shape == (H, W, 3)
    assert np.array_equal(a, b), '%d of %d bytes differ, max %d' % (int((a != b).sum()), a.size, int(np.abs(a.astype(int) - b.astype(int)).max()))


def test_the_default_windows_of_the_frame_loop(oracle, cvr):
    """The crops process_kenburns actually asks for (kbe.py:130-140: 0.90 / 0.85 of the frame, KBE; 0.8 / 0.3, dolly)."""
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from ken_burns_effect_amd import common, synthetic
    rng = np.random.default_rng(3)
    for H, W in ((256, 256), (300, 400), (512, 512)):
        frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        for dolly in (False, True):
            ofrom, oto = synthetic.default_windows(H, W, dolly)
            cw, ch = common.crop_size({'objectFrom': ofrom, 'objectTo': oto})
            assert np.array_equal(oracle.crop_resize_u8(frame, cw, ch), _second(cvr, frame, cw, ch)), (H, W, dolly)


def test_window_sticking_out_of_the_image_replicates_the_border(cvr):
    """getRectSubPix on its own, centre near a corner (the frame loop never does this; OpenCV replicates the border)."""
    rng = np.random.default_rng(9)
    src = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    dst = np.empty((9, 11, 3), np.uint8)
    cvr.cvr_get_rect_sub_pix(src.ctypes.data_as(ctypes.c_void_p), 30, 20, dst.ctypes.data_as(ctypes.c_void_p), 11, 9, ctypes.c_float(2.0), ctypes.c_float(1.0))
    # integer centre, odd window: whole-pixel sampling -> a replicate-padded copy
    pad = np.pad(src, ((10, 10), (10, 10), (0, 0)), mode='edge')
    want = pad[10 + 1 - 4:10 + 1 + 5, 10 + 2 - 5:10 + 2 + 6]
    assert np.array_equal(dst, want)


def test_geometry_against_pillows_bilinear_resize(oracle):
    """A third party's view of the GEOMETRY (not of OpenCV's fixed-point arithmetic): Pillow's bilinear resize of the box getRectSubPix
    samples -- patch pixel i is the source position W/2 - (cw - 1)/2 + i in pixel-centre coordinates (common.py:256), i.e. the box
    [x0, x0 + cw] in Pillow's pixel-edge coordinates -- back to W x H.  Where the window starts on a whole pixel the patch is a copy
    and both are ONE bilinear interpolation on the half-pixel grid: within one count everywhere inside the border, on NOISE (a
    convention that is off by half a pixel reads 40 counts off on average).  Where it starts on a half pixel OpenCV interpolates
    twice (getRectSubPix rounds a 2 x 2 blend to bytes, resize blends those): close on a smooth image only.  The outermost pixels are
    left out: Pillow reads the source beyond the box there, OpenCV replicates the patch's edge."""
    Image = pytest.importorskip('PIL.Image')
    rng = np.random.default_rng(1)

    def smooth(H, W):
        y, x = np.mgrid[0:H, 0:W].astype(np.float64)
        out = np.zeros((H, W, 3))
        for c in range(3):
            for _ in range(4):
                fx, fy, ph = rng.uniform(0.002, 0.03), rng.uniform(0.002, 0.03), rng.uniform(0, 6.28)
                out[..., c] += np.sin(6.28 * (x * fx + y * fy) + ph)
        return ((out - out.min()) / (out.max() - out.min()) * 255).round().astype(np.uint8)

    def pillow(frame, cw, ch, off=0.0):
        H, W, _ = frame.shape
        x0, y0 = W / 2.0 - (cw - 1) * 0.5 + off, H / 2.0 - (ch - 1) * 0.5 + off
        return np.asarray(Image.fromarray(frame).resize((W, H), Image.BILINEAR, box=(x0, y0, x0 + cw, y0 + ch)))

    for (H, W, cw, ch) in [(256, 256, 231, 231), (300, 400, 361, 271), (512, 512, 461, 461), (1024, 1024, 921, 921), (96, 128, 39, 29)]:
        frame = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)                 # whole-pixel windows, noise
        ours = oracle.crop_resize_u8(frame, cw, ch).astype(np.int32)
        d = np.abs(ours - pillow(frame, cw, ch))[3:-3, 3:-3]
        assert d.max() <= 1 and d.mean() < 0.3, ((H, W, cw, ch), int(d.max()), float(d.mean()))
        off = np.abs(ours - pillow(frame, cw, ch, 0.5))[3:-3, 3:-3]                 # ... and the test can tell: half a pixel off
        assert off.mean() > 20, ((H, W, cw, ch), float(off.mean()))
    for (H, W, cw, ch) in [(256, 256, 230, 230), (300, 400, 360, 270), (1024, 1024, 920, 920)]:
        frame = smooth(H, W)                                                    # half-pixel windows, a smooth image
        ours = oracle.crop_resize_u8(frame, cw, ch).astype(np.int32)
        d = np.abs(ours - pillow(frame, cw, ch))[3:-3, 3:-3]
        assert d.max() <= 2 and d.mean() < 0.4, ((H, W, cw, ch), int(d.max()), float(d.mean()))
        off = np.abs(ours - pillow(frame, cw, ch, 0.5))[3:-3, 3:-3]
        assert off.mean() > 2, ((H, W, cw, ch), float(off.mean()))


def test_against_cv2_itself_when_opencv_is_installed(oracle, cvr):
    """THE pin of SURVEY 8f row 1: cv2.getRectSubPix + cv2.resize called as /root/reference/utils/common.py:256-257 calls them, on
    the crops process_kenburns asks for (kbe.py:130-140) at 512^2 and 1024^2 (and an odd size), against the oracle's restatement
    -- what the HIP kernel k_crop_resize_u8 is tested against byte for byte -- and the second restatement.  Skipped where OpenCV is
    absent (this image); the first image with OpenCV >= 4 turns the row from "parity unpinned" to pinned, or names the bytes."""
    cv2 = pytest.importorskip('cv2')
    import sys
    sys.path.insert(0, os.path.dirname(HERE))
    from ken_burns_effect_amd import common, synthetic
    rng = np.random.default_rng(11)
    for H, W in ((512, 512), (1024, 1024), (300, 400)):
        noise = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        yy, xx = np.mgrid[0:H, 0:W]
        smooth = np.stack([(xx * 255 // max(W - 1, 1)), (yy * 255 // max(H - 1, 1)), ((xx + yy) * 255 // max(W + H - 2, 1))], -1).astype(np.uint8)
        for frame in (noise, smooth):
            for dolly in (False, True):
                ofrom, oto = synthetic.default_windows(H, W, dolly)
                cw, ch = common.crop_size({'objectFrom': ofrom, 'objectTo': oto})
                want = cv2.getRectSubPix(image=frame, patchSize=(cw, ch), center=(W / 2.0, H / 2.0))
                want = cv2.resize(src=want, dsize=(W, H), fx=0.0, fy=0.0, interpolation=cv2.INTER_LINEAR)
                for name, got in (('oracle.crop_resize_u8', oracle.crop_resize_u8(frame, cw, ch)), ('opencv_8u_restatement.c', _second(cvr, np.ascontiguousarray(frame), cw, ch))):
                    d = np.abs(got.astype(np.int32) - want.astype(np.int32))
                    assert np.array_equal(got, want), '%s vs cv2 %s at %dx%d crop %dx%d: %d of %d bytes differ, max %d' % (name, cv2.__version__, W, H, cw, ch, int((d > 0).sum()), d.size, int(d.max()))


def test_min_max_loc_against_cv2_when_opencv_is_installed():
    """cv2.minMaxLoc as /root/reference/utils/pipeline.py:96 calls it (the border-cropped depth map -> objectDepthrange) against
    synthetic.depthrange_of: values and the tie-breaking of the two locations (first in row-major order).  Skipped without OpenCV."""
    cv2 = pytest.importorskip('cv2')
    import sys
    import torch
    sys.path.insert(0, os.path.dirname(HERE))
    from ken_burns_effect_amd import synthetic
    rng = np.random.default_rng(12)
    for H, W in ((300, 400), (512, 512)):
        depth = rng.random((H, W), dtype=np.float32) * 100 + 1
        depth[140, 150] = depth[200, 170] = 0.5          # the minimum, twice
        depth[150, 140] = depth[150, 260] = 500.0        # the maximum, twice
        want = cv2.minMaxLoc(src=depth[128:-128, 128:-128], mask=None)
        got = synthetic.depthrange_of(torch.from_numpy(depth).view(1, 1, H, W))
        assert (float(got[0]), float(got[1]), tuple(got[2]), tuple(got[3])) == (float(want[0]), float(want[1]), tuple(want[2]), tuple(want[3]))
