"""Cross-check of the OpenCV 8-bit paths of common.py:256-257 (cv2.getRectSubPix + cv2.resize INTER_LINEAR): the oracle's
numpy restatement (what the HIP kernel k_crop_resize_u8 is tested against, tests/test_hip_parity.py) versus a second,
independently written restatement that follows the structure of OpenCV's own implementation
(tests/opencv_8u_restatement.c).  OpenCV is not installed here: agreement of two restatements catches transcription
errors, it does not pin parity with cv2 -- DESIGN.md keeps that row "parity unpinned"."""
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
